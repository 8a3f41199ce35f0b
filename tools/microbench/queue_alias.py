"""HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4).  Does the pipelined step survive when the
process has created other streams before the compute stream and the library's side stream (as torch.distributed's NCCL
streams would be at N > 1)?  Creates `ndummy` idle streams first, then times the bench loop (no collective)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from sp_orb_slam_amd import parallel, synth, weights
from sp_orb_slam_amd.extractor import SPExtractor

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
H, W, nf, B = 480, 752, 1000, 8
blob = weights.synthetic(7, "dense")
d_img = torch.from_numpy(np.stack([synth.make_image(100 + i, H, W) for i in range(B)])).cuda()
keep = []
for ndummy in (0, 1, 2, 3, 4, 5, 6):
    keep.append([torch.cuda.Stream() for _ in range(ndummy)])
    for s_ in keep[-1]:
        with torch.cuda.stream(s_):
            torch.zeros(1, device="cuda")      # make sure the stream really exists
    stream = torch.cuda.Stream()
    ext = SPExtractor(nf, H, W, blob, max_batch=B, precision=prec, with_heat=False, async_cov=True)
    sh = parallel.ShardedExtractor(ext, 1, 0, B)
    for _ in range(5):
        sh.step(d_img, stream)
    sh.flush(stream); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        sh.step(d_img, stream)
    sh.flush(stream); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print("%s GPU_MAX_HW_QUEUES=%s extra streams so far %d: %.4f ms per step" % (prec, os.environ.get("GPU_MAX_HW_QUEUES", "default"), sum(len(k) for k in keep), dt * 1e3))
    ext.close()
