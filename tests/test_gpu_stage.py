"""GPU: input staging (spfe_set_staging / spfe_extract_staged / spfe_stage_batch_device) vs the
oracle's restatement of cv::remap + crop + cvtColor — bit-exact gray frames, and the staged
extraction equal to extracting the oracle's gray frame.  SURVEY.md §8(f) rank 2."""
import numpy as np
import pytest

from oracle import oracle
from sp_orb_slam_amd import weights
from sp_orb_slam_amd.extractor import SPExtractor, SpfeError

pytestmark = pytest.mark.gpu


def undistort_maps(hs, ws, k1=-0.28340811, k2=0.07395907, p1=0.00019359, p2=1.76187114e-05):
    """Maps of the kind cv::initUndistortRectifyMap produces for the EuRoC intrinsics hard-coded at
    data_loader.cc:471-481 (scaled to the test size); computed in float64, stored f32."""
    fx, fy, cx, cy = 458.654 * ws / 752, 457.296 * hs / 480, 367.215 * ws / 752, 248.375 * hs / 480
    v, u = np.mgrid[0:hs, 0:ws].astype(np.float64)
    x, y = (u - cx) / fx, (v - cy) / fy
    r2 = x * x + y * y
    kr = 1 + k1 * r2 + k2 * r2 * r2
    xd = x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return (xd * fx + cx).astype(np.float32), (yd * fy + cy).astype(np.float32)


def _raw(rng, hs, ws, cn):
    base = rng.integers(0, 256, (hs // 4 + 1, ws // 4 + 1, cn)).astype(np.float32)
    up = np.kron(base, np.ones((4, 4, 1), np.float32))[:hs, :ws]
    img = np.clip(up + rng.normal(0, 12, (hs, ws, cn)), 0, 255).astype(np.uint8)
    return img[..., 0] if cn == 1 else img


@pytest.mark.parametrize("cn,rgb,with_map,hs,ws", [(3, False, True, 120, 160), (3, True, True, 136, 200),
                                                   (1, False, True, 120, 160), (4, False, True, 128, 168),
                                                   (4, True, False, 120, 160), (3, False, False, 150, 180)])
def test_staged_gray_and_extraction_match_oracle(cn, rgb, with_map, hs, ws):
    H, W, nf = 120, 160, 100
    rng = np.random.default_rng(cn * 100 + hs)
    blob = weights.synthetic(7, "dense")
    ext = SPExtractor(nf, H, W, blob, max_batch=2)
    mx, my = undistort_maps(hs, ws) if with_map else (None, None)
    if with_map:   # also exercise the borders: shift part of the map outside the source
        mx = mx.copy(); my = my.copy()
        mx[:, :5] -= 7.3
        my[-4:, :] += 9.9
    ext.set_staging(hs, ws, cn, rgb, mx, my)
    raws = [_raw(rng, hs, ws, cn) for _ in range(2)]
    frs = ext.extract_batch_staged(raws)
    for i, (raw, fr) in enumerate(zip(raws, frs)):
        want = oracle.stage_input(raw, H, W, mx, my, rgb)
        assert np.array_equal(ext.debug_read("image", i), want)
        ref = oracle.extract(blob, want, nf)
        assert fr.K == ref["K"] and np.array_equal(fr.kp_xy, ref["kp_xy"])
        assert np.array_equal(fr.descriptors.view(np.uint32), ref["desc"].view(np.uint32))
        assert np.array_equal(fr.cov2_inv.view(np.uint32), ref["cov2_inv"].view(np.uint32))
    one = ext.extract_staged(raws[1])
    assert one.K == frs[1].K and np.array_equal(one.descriptors, frs[1].descriptors)
    ext.close()


def test_staging_full_size_752x480_bgr():
    H, W, nf = 480, 752, 1000
    rng = np.random.default_rng(8)
    blob = weights.synthetic(7, "sparse")
    ext = SPExtractor(nf, H, W, blob, with_heat=False)
    mx, my = undistort_maps(H, W)
    ext.set_staging(H, W, 3, False, mx, my)
    raw = _raw(rng, H, W, 3)
    fr = ext.extract_staged(raw)
    want = oracle.stage_input(raw, H, W, mx, my)
    assert np.array_equal(ext.debug_read("image"), want)
    ref = oracle.extract(blob, want, nf)
    assert fr.K == ref["K"] and np.array_equal(fr.kp_xy, ref["kp_xy"]) and np.array_equal(fr.occ_grid, ref["occ_grid"])
    ext.close()


def test_stage_device_then_extract_device():
    import torch

    H, W, nf, B, hs, ws = 120, 160, 100, 3, 128, 176
    rng = np.random.default_rng(4)
    blob = weights.synthetic(7, "dense")
    ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False)
    mx, my = undistort_maps(hs, ws)
    ext.set_staging(hs, ws, 3, False, mx, my)
    raws = np.stack([_raw(rng, hs, ws, 3) for _ in range(B)])
    d_raw = torch.from_numpy(raws).cuda()
    d_gray = torch.zeros((B, H, W), dtype=torch.uint8, device="cuda")
    rec = torch.zeros(B * ext.record_bytes(), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    ext.stage_batch_device(d_raw.data_ptr(), B, d_gray.data_ptr(), s.cuda_stream)
    ext.extract_batch_device(d_gray.data_ptr(), B, rec.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    g = d_gray.cpu().numpy()
    rb = ext.record_bytes()
    hrec = rec.cpu().numpy()
    for i in range(B):
        want = oracle.stage_input(raws[i], H, W, mx, my)
        assert np.array_equal(g[i], want)
        fr = ext.view_record(hrec[i * rb:(i + 1) * rb])
        ref = oracle.extract(blob, want, nf)
        assert fr.K == ref["K"] and np.array_equal(fr.kp_xy, ref["kp_xy"])
    ext.close()


def test_staging_errors():
    ext = SPExtractor(50, 64, 96, weights.synthetic(7, "dense"))
    raw = np.zeros((64, 96, 3), np.uint8)
    with pytest.raises(SpfeError, match="spfe_set_staging"):
        ext._staging = (64, 96, 3)
        ext.extract_staged(raw)
    with pytest.raises(SpfeError, match="smaller"):
        ext.set_staging(60, 96, 3)
    with pytest.raises(SpfeError, match="channels"):
        ext.set_staging(64, 96, 2)
    ext.set_staging(64, 96, 3)
    with pytest.raises(RuntimeError, match="input image is empty"):
        ext.extract_staged(None)
    with pytest.raises(SpfeError, match="raw frame"):
        ext.extract_staged(np.zeros((64, 96), np.uint8))
    ext.close()
