"""ATen-CPU op sequence of the SuperPoint extraction network.

The SAME op sequence as SPFrontend::forward (/root/reference/orb_slam2/src/cv/
sp_extractor.cpp:79-159) plus the input conversion (:388), issued on PyTorch-CPU:
conv2d/relu/max_pool2d/softmax/max/gather/masked_select/clamp/log/pixel_shuffle/
grid_sampler_2d/norm.  This is the builder's restatement, not reference code (the
reference hard-wires CUDA, :73,:134,:348-351, and cannot run on a CPU).

Two users, neither in the product path:
  * tests/golden/make_golden.py (build container): fixtures that pin the C oracle;
  * bench.py `cpu_baseline_aten`: "the reference's CPU libtorch path timed on the same
    box's host cores" of BASELINE.json's north_star — the closest thing that can exist,
    since libtorch-CPU == ATen-CPU and torch is on the GPU box for device plumbing.
"""
import numpy as np
import torch
import torch.nn.functional as F


def forward(named, img_u8, cuda_scalar_div=False):
    """SPFrontend::forward (:79-159) + input conversion (:388), ATen-CPU, NCHW.
    cuda_scalar_div: evaluate `pixels.div(W / 2.0)` (:137-138) the way libtorch-1.6's CUDA kernel does — tensor / scalar is
    `a * inv_b` with inv_b = accscalar_t(1.0) / b formed once in f32 (aten/src/ATen/native/cuda/BinaryMulDivKernel.cu,
    div_kernel_cuda's is_cpu_scalar branch) — instead of ATen-CPU's true division.  The reference hard-wires CUDA (:73), so
    this is the form the oracle and the kernels pin; the two differ in the last bit of some sampling coordinates."""
    H, W = img_u8.shape
    hc, wc = H // 8, W // 8
    t = {k: torch.from_numpy(v) for k, v in named.items()}
    x = torch.from_numpy(img_u8.astype(np.float32) * np.float32(1.0 / 255.0))[None, None]

    def conv(name, x, pad):
        return F.conv2d(x, t[name + ".weight"], t[name + ".bias"], stride=1, padding=pad)

    x = torch.relu(conv("conv1a", x, 1))
    x = torch.relu(conv("conv1b", x, 1))
    x = F.max_pool2d(x, 2, 2)
    x = torch.relu(conv("conv2a", x, 1))
    x = torch.relu(conv("conv2b", x, 1))
    x = F.max_pool2d(x, 2, 2)
    x = torch.relu(conv("conv3a", x, 1))
    x = torch.relu(conv("conv3b", x, 1))
    x = F.max_pool2d(x, 2, 2)
    x = torch.relu(conv("conv4a", x, 1))
    x = torch.relu(conv("conv4b", x, 1))
    cPa = torch.relu(conv("convPa", x, 1))
    semi = conv("convPb", cPa, 0).squeeze()
    cDa = torch.relu(conv("convDa", x, 1))
    coarse_raw = conv("convDb", cDa, 0)
    dn = torch.norm(coarse_raw, 2, 1)
    coarse = coarse_raw.div(torch.unsqueeze(dn, 1))

    dense = torch.softmax(semi, 0)
    semi_dust = semi[-1]
    dense_dust = dense[-1]
    nodust = dense[:-1]
    score, indices = nodust.max(0)

    # `grid` of the SPFrontend ctor (:64-73)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    grid_ = torch.cat([xx.unsqueeze(0), yy.unsqueeze(0)])
    grid = (grid_.contiguous().view(1, 2, H // 8, 8, W // 8, 8)
            .permute(0, 1, 3, 5, 2, 4).reshape(1, 2, 64, hc, wc))
    idx = indices.view(1, 1, 1, hc, wc).expand(-1, 2, -1, -1, -1)
    pixel = torch.gather(grid, 2, idx)

    mask = score >= 0.007
    pixels_in = torch.masked_select(pixel, mask).reshape(2, -1).type_as(semi)
    score_sel = torch.masked_select(score, mask)

    heat_log = F.pixel_shuffle(torch.log(torch.clamp(nodust, 0.001)).unsqueeze(0), 8)

    if cuda_scalar_div:
        x_s = pixels_in[0] * float(np.float32(1.0) / np.float32(W / 2.0)) - 1.0
        y_s = pixels_in[1] * float(np.float32(1.0) / np.float32(H / 2.0)) - 1.0
    else:
        x_s = pixels_in[0].div(W / 2.0) - 1.0
        y_s = pixels_in[1].div(H / 2.0) - 1.0
    samp = torch.cat([x_s.unsqueeze(-1), y_s.unsqueeze(-1)], -1).unsqueeze(0).unsqueeze(0)
    n = pixels_in.shape[1]
    if n > 0:
        desc = torch.grid_sampler_2d(coarse, samp, 0, 0, True).squeeze(2).squeeze(0)  # [256, N]
        desc = desc.div(torch.norm(desc, 2, 0, True))
    else:
        desc = torch.zeros(256, 0)
    return dict(semi=semi.permute(1, 2, 0).contiguous().numpy(),          # [hc,wc,65]
                coarse_raw=coarse_raw[0].permute(1, 2, 0).contiguous().numpy(),  # [hc,wc,256]
                semi_dust=semi_dust.numpy().copy(), dense_dust=dense_dust.numpy().copy(),
                pixels_in=pixels_in.numpy().copy(), score=score_sel.numpy().copy(),
                score_map=score.numpy().copy(), argmax_map=indices.numpy().astype(np.int32),
                desc=desc.numpy().T.copy(), heat_log=heat_log[0, 0].numpy().copy())


def time_forward(named, images, threads, seconds=3.0, max_frames=64):
    """frames/s of forward() over `images` with torch.set_num_threads(threads);
    one untimed warm-up frame.  Returns (fps, frames, elapsed)."""
    import time
    torch.set_num_threads(threads)
    with torch.no_grad():
        forward(named, images[0])
        done, t0 = 0, time.perf_counter()
        while True:
            forward(named, images[done % len(images)])
            done += 1
            el = time.perf_counter() - t0
            if el >= seconds or done >= max_frames:
                break
    return done / el, done, el
