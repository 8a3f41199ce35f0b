"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through
the C ABI (sp_orb_slam_amd.extractor -> libspfe.so), against the CPU oracle on
the same seeded inputs, and against the committed ATen-CPU golden fixtures.

Bars:
  * keypoint positions / order, occ_grid, candidate count: bit-exact;
  * descriptors: max-abs <= 2e-5 on unit vectors (f32);   DESC_TOL
  * heat / heat_inv / response: abs <= 1e-5;               HEAT_TOL
  * cov2 / cov2_inv: rel <= 1e-5.                          COV_RTOL
The f32 MFMA path follows the same operation order as the oracle
(include/spfe_exact_math.h), so the tests additionally report (and for the
network outputs require) BITWISE equality.
"""
import os

import numpy as np
import pytest

from oracle import oracle
from sp_orb_slam_amd import synth, weights
from sp_orb_slam_amd.extractor import SPExtractor, SpfeError, math_probe

pytestmark = pytest.mark.gpu

DESC_TOL = 2e-5
HEAT_TOL = 1e-5
COV_RTOL = 1e-5


def _compare(fr, ref, with_heat=True):
    assert fr.n_candidates == ref["n_candidates"]
    assert fr.K == ref["K"]
    assert np.array_equal(fr.kp_xy, ref["kp_xy"])
    assert np.array_equal(fr.occ_grid, ref["occ_grid"])
    assert np.abs(fr.descriptors - ref["desc"]).max(initial=0) <= DESC_TOL
    assert np.abs(fr.response - ref["response"]).max(initial=0) <= HEAT_TOL
    assert np.allclose(fr.cov2, ref["cov2"], rtol=COV_RTOL, atol=0)
    assert np.allclose(fr.cov2_inv, ref["cov2_inv"], rtol=COV_RTOL, atol=0)
    assert np.abs(fr.dense_dust - ref["dense_dust"]).max() <= 1e-6
    assert np.allclose(fr.semi_dust, ref["semi_dust"], rtol=1e-6, atol=1e-6)
    if with_heat:
        assert np.allclose(fr.heat, ref["heat"], rtol=0, atol=HEAT_TOL, equal_nan=True)
        assert np.allclose(fr.heat_inv, ref["heat_inv"], rtol=0, atol=HEAT_TOL, equal_nan=True)


def test_exact_math_device_bits_equal_host_bits():
    rng = np.random.default_rng(0)
    x = np.concatenate([-rng.random(4096).astype(np.float32) * 90.0,
                        -np.float32(10.0) ** rng.uniform(-6, 1.9, 4096).astype(np.float32),
                        np.array([0.0, -0.0, -86.0, -86.5, -1e-30], np.float32)])
    e, _ = math_probe(x)
    ref_e = np.array([oracle.lib().oracle_expf(float(v)) for v in x], np.float32)
    assert np.array_equal(e.view(np.uint32), ref_e.view(np.uint32))
    y = np.concatenate([rng.uniform(0.001, 1.0, 8192).astype(np.float32),
                        np.array([0.001, 1.0, 0.5, 0.70710677, 0.0153846], np.float32)])
    _, l = math_probe(y)
    ref_l = np.array([oracle.lib().oracle_logf(float(v)) for v in y], np.float32)
    assert np.array_equal(l.view(np.uint32), ref_l.view(np.uint32))
    # and both are accurate
    assert np.abs(ref_l - np.log(y.astype(np.float64))).max() < 3e-7
    assert (np.abs(ref_e - np.exp(x.astype(np.float64))) / np.exp(x.astype(np.float64)))[x > -80].max() < 3e-7


@pytest.mark.parametrize("H,W,iseed,wseed,det,nf", [
    (64, 96, 1, 7, "dense", 20),
    (64, 96, 2, 7, "sparse", 100),
    (128, 160, 5, 11, "sparse", 200),
    (72, 104, 9, 7, "dense", 1000),     # tiles partially outside the image at every level
    (240, 320, 3, 7, "dense", 300),
    (480, 640, 1, 7, "sparse", 1000),   # BASELINE configs[0] shape
])
def test_full_path_matches_oracle(H, W, iseed, wseed, det, nf):
    img = synth.make_image(iseed, H, W)
    blob = weights.synthetic(wseed, det)
    ext = SPExtractor(nf, H, W, blob)
    kps, desc = ext(img, None)
    ref = oracle.extract(blob, img, nf)
    semi = ext.debug_read("semi")
    coarse = ext.debug_read("coarse")
    # network: bitwise (same fma order as the oracle)
    assert np.array_equal(semi.view(np.uint32), ref["semi"].view(np.uint32))
    assert np.array_equal(coarse.view(np.uint32), ref["coarse"].view(np.uint32))
    _compare(ext.last, ref)
    assert np.array_equal(np.stack([kps["x"], kps["y"]], 1), ref["kp_xy"])
    assert desc.shape == (ref["K"], 256) and desc.dtype == np.float32
    assert np.all(kps["size"] == 1.0) and np.all(kps["angle"] == -1.0) and np.all(kps["octave"] == 0)
    assert np.array_equal(kps["response"], ext.last.response)
    ext.close()


@pytest.mark.parametrize("H,W", [(16, 16), (16, 40), (40, 16), (24, 32), (32, 24), (56, 72), (104, 40), (48, 264)])
def test_odd_sizes_exact(H, W):
    """Small and lopsided sizes: fewer work items than persistent workgroups, tiles that are
    mostly outside the image, 2x2-cell grids."""
    blob = weights.synthetic(7, "dense")
    img = synth.make_image(H + W, H, W)
    ext = SPExtractor(64, H, W, blob, max_batch=3)
    ref = oracle.extract(blob, img, 64)
    for fr in ext.extract_batch([img, img, img]):
        _compare(fr, ref)
    assert np.array_equal(ext.debug_read("coarse", 2).view(np.uint32), ref["coarse"].view(np.uint32))
    ext.close()


def test_layerwise_activations_bitwise():
    """Every conv stage against the oracle's network on a size with ragged tiles."""
    H, W = 88, 120
    img = synth.make_image(4, H, W)
    blob = weights.synthetic(7, "dense")
    ext = SPExtractor(50, H, W, blob)
    ext(img, None)
    _, _, feat = oracle.network(blob, img)
    got = ext.debug_read("feat")
    assert np.array_equal(got.view(np.uint32), feat.view(np.uint32))
    ext.close()


@pytest.mark.parametrize("H,W,B", [(88, 120, 1), (480, 752, 2), (120, 160, 8), (64, 96, 3)])
def test_fused_first_layer_equals_separate_conv1a(H, W, B, monkeypatch):
    """conv1a computed inside conv1b (SPFE_FUSE_CONV1A=1) vs the separate conv1a kernel (default):
    conv1b's output and everything after it bit-identical, over sizes with ragged tiles, several
    tiles per workgroup (the image patch double buffer) and batches."""
    blob = weights.synthetic(7, "dense")
    imgs = [synth.make_image(40 + i, H, W) for i in range(B)]
    outs = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("SPFE_FUSE_CONV1A", fuse)
        ext = SPExtractor(100, H, W, blob, max_batch=B, with_heat=False)
        frs = ext.extract_batch(imgs)
        act1 = [ext.debug_read("act1", i) for i in range(B)]
        if fuse == "1":
            with pytest.raises(Exception, match="not materialised"):
                ext.debug_read("act0")
        outs.append((frs, act1))
        ext.close()
    for i in range(B):
        assert np.array_equal(outs[0][1][i].view(np.uint32), outs[1][1][i].view(np.uint32))
        a, b = outs[0][0][i], outs[1][0][i]
        assert a.K == b.K and np.array_equal(a.kp_xy, b.kp_xy)
        assert np.array_equal(a.descriptors.view(np.uint32), b.descriptors.view(np.uint32))
    ref = oracle.extract(blob, imgs[-1], 100)
    assert outs[0][0][-1].K == ref["K"] and np.array_equal(outs[0][0][-1].kp_xy, ref["kp_xy"])


def test_base_extractor_getters():
    """BaseExtractor with ctor args (n, 1.0, 1, 1, 1) (sp_extractor.cpp:343): one level, unit scales —
    what Frame reads at frame.cpp:211-217."""
    ext = SPExtractor(10, 64, 96, weights.synthetic(7, "dense"))
    assert ext.GetLevels() == 1 and ext.GetScaleFactor() == 1.0
    assert ext.GetScaleFactors() == [1.0] and ext.GetInverseScaleFactors() == [1.0]
    assert ext.GetScaleSigmaSquares() == [1.0] and ext.GetInverseScaleSigmaSquares() == [1.0]
    ext.close()


def test_bench_config_752x480_1000_keypoints():
    """BASELINE configs[1]: 752x480, 1000 keypoints, f32."""
    H, W, nf = 480, 752, 1000
    blob = weights.synthetic(7, "dense")
    ext = SPExtractor(nf, H, W, blob)
    for seed in (100, 101):
        img = synth.make_image(seed, H, W)
        ext(img, None)
        ref = oracle.extract(blob, img, nf)
        _compare(ext.last, ref)
        assert ext.last.K <= nf + 1 and ext.last.K > 900
    ext.close()


@pytest.mark.parametrize("name", ["g64x96_dense", "g64x96_sparse", "g128x160_sparse",
                                  "g480x752_dense", "g480x640_sparse", "g720x1280_sparse"])
def test_against_aten_golden(name, golden_dir):
    """HIP path vs the fixtures generated by the ATen-CPU op sequence."""
    g = np.load("%s/%s.npz" % (golden_dir, name))
    H, W, nf = int(g["meta_H"]), int(g["meta_W"]), int(g["meta_num_features"])
    img = g["image"] if "image" in g.files else synth.make_image(int(g["meta_image_seed"]), H, W)
    blob = weights.synthetic(int(g["meta_weight_seed"]), str(g["meta_detector"]))
    ext = SPExtractor(nf, H, W, blob)
    ext(img, None)
    fr = ext.last
    assert fr.n_candidates == int(g["n_candidates"])
    assert np.array_equal(fr.kp_xy.astype(np.int16), g["kp_xy"])
    assert np.array_equal(fr.occ_grid, g["occ_grid"])
    assert np.abs(fr.response - g["response"]).max() <= HEAT_TOL
    assert np.allclose(fr.cov2_inv, g["cov2_inv"], rtol=COV_RTOL)
    assert np.abs(fr.dense_dust - g["dense_dust"]).max() <= 1e-5
    if "kp_desc" in g.files:
        assert np.abs(fr.descriptors - g["kp_desc"]).max() <= DESC_TOL
        assert np.abs(fr.heat - g["heat"]).max() <= HEAT_TOL
    else:
        assert np.abs(fr.descriptors[::16] - g["kp_desc_sub"]).max() <= DESC_TOL
        assert np.abs(fr.heat[H // 2] - g["heat_row"]).max() <= HEAT_TOL
    ext.close()


def test_batch_equals_single_frames():
    H, W, nf, n = 120, 160, 150, 5
    blob = weights.synthetic(7, "dense")
    imgs = [synth.make_image(30 + i, H, W) for i in range(n)]
    ext1 = SPExtractor(nf, H, W, blob)
    single = []
    for im in imgs:
        ext1(im, None)
        single.append(ext1.last)
    extb = SPExtractor(nf, H, W, blob, max_batch=8)
    batch = extb.extract_batch(imgs)
    for a, b in zip(single, batch):
        assert a.K == b.K and np.array_equal(a.kp_xy, b.kp_xy)
        assert np.array_equal(a.descriptors, b.descriptors)  # same kernels, same order: bitwise
        assert np.array_equal(a.occ_grid, b.occ_grid)
        assert np.array_equal(a.cov2_inv, b.cov2_inv)
        assert np.array_equal(a.heat, b.heat)
    ext1.close()
    extb.close()


def test_strided_input_and_repeat_calls():
    H, W, nf = 64, 96, 50
    blob = weights.synthetic(7, "dense")
    big = np.zeros((H, W + 32), np.uint8)
    img = synth.make_image(8, H, W)
    big[:, :W] = img
    ext = SPExtractor(nf, H, W, blob)
    k1, d1 = ext(big[:, :W], None)   # row stride W+32 (cv::Mat::step)
    k2, d2 = ext(img, None)
    assert np.array_equal(k1, k2) and np.array_equal(d1, d2)
    ext.close()


def test_error_behaviour():
    H, W = 64, 96
    blob = weights.synthetic(7, "dense")
    ext = SPExtractor(10, H, W, blob)
    with pytest.raises(RuntimeError, match="input image is empty"):  # sp_extractor.cpp:364-365
        ext(np.zeros((0, 0), np.uint8), None)
    with pytest.raises(RuntimeError, match="input image is empty"):
        ext(None, None)
    with pytest.raises(SpfeError):
        ext(np.zeros((H, W + 8), np.uint8), None)
    with pytest.raises(SpfeError):
        ext(np.zeros((H, W), np.float32), None)
    with pytest.raises(SpfeError):
        ext.extract_batch([np.zeros((H, W), np.uint8)] * 2)  # max_batch = 1
    ext.close()


def test_zero_and_few_candidates():
    """N = 0 and very small N (the reference's squeeze() at :146 mis-shapes N = 1; here every N is defined)."""
    H, W = 64, 96
    named = weights.to_named_tensors(weights.synthetic(7, "dense"))
    named["convPb.weight"][:] = 0
    named["convPb.bias"][:] = 0
    named["convPb.bias"][64] = 30.0  # all mass on the dustbin: no cell reaches 0.007
    blob0 = weights.from_named_tensors(named)
    img = synth.make_image(1, H, W)
    ext = SPExtractor(10, H, W, blob0)
    kps, desc = ext(img, None)
    ref = oracle.extract(blob0, img, 10)
    assert ref["K"] == 0 and len(kps) == 0 and desc.shape == (0, 256)
    assert np.all(ext.occ_grid_ == -1) and ext.last.n_candidates == 0
    # every probability is clamped to 0.001 -> the heat map is constant -> to_heat divides
    # by (max - min) == 0 exactly as the reference does (:467-468): NaN on both sides
    assert np.isnan(ext.heat_).all() and np.isnan(ref["heat"]).all()
    ext.close()
    # a handful of candidates: strong dustbin, a few cells still pass
    named = weights.to_named_tensors(weights.synthetic(7, "dense"))
    named["convPb.weight"] *= np.float32(3.5)
    for db in (8.6, 8.3, 8.0):
        named["convPb.bias"][64] = np.float32(db)
        blob1 = weights.from_named_tensors(named)
        ref = oracle.extract(blob1, img, 10)
        ext = SPExtractor(10, H, W, blob1)
        ext(img, None)
        _compare(ext.last, ref)
        ext.close()


def test_1280x720_batch_matches_oracle():
    """BASELINE configs[3] shape (1280x720), f32 path, batch of 2: exact against the oracle."""
    H, W, nf = 720, 1280, 800
    blob = weights.synthetic(7, "sparse")
    imgs = [synth.make_image(300 + i, H, W) for i in range(2)]
    ext = SPExtractor(nf, H, W, blob, max_batch=2)
    for fr, im in zip(ext.extract_batch(imgs), imgs):
        _compare(fr, oracle.extract(blob, im, nf))
    ext.close()


def test_nms_cut_and_border_properties_720p():
    """1280x720 (BASELINE configs[3] shape): size-independent properties of the selection."""
    H, W, nf = 720, 1280, 1000
    blob = weights.synthetic(7, "dense")
    img = synth.make_image(300, H, W)
    ext = SPExtractor(nf, H, W, blob, with_heat=False)
    kps, desc = ext(img, None)
    fr = ext.last
    assert 0 < fr.K <= nf + 1
    x, y = fr.kp_xy[:, 0].astype(int), fr.kp_xy[:, 1].astype(int)
    assert np.all((x >= 8) & (x < W - 8) & (y >= 8) & (y < H - 8))          # border 8
    key = y * W + x
    assert np.all(np.diff(key) > 0)                                          # raster order
    cells = (y // 8) * (W // 8) + x // 8
    assert len(np.unique(cells)) == fr.K                                     # <= 1 kp per cell
    occ = fr.occ_grid.reshape(-1)
    assert np.array_equal(np.sort(occ[occ >= 0]), np.arange(fr.K))
    assert np.array_equal(occ[cells], np.arange(fr.K))
    dx = np.abs(x[:, None] - x[None, :]); dy = np.abs(y[:, None] - y[None, :])
    close = (np.maximum(dx, dy) <= 4) & ~np.eye(fr.K, dtype=bool)
    assert not close.any()                                                   # NMS radius 4
    assert np.abs(np.linalg.norm(desc, axis=1) - 1).max() < 1e-6             # unit descriptors
    assert np.all(fr.cov2 >= 1.0) and np.allclose(fr.cov2 * fr.cov2_inv, 1, rtol=1e-6)
    ext.close()


def test_cpp_adaptor_matches_python_path(tmp_path):
    """The C++ adaptor (include/spfe_extractor.hpp), driven like Frame::ExtractORB drives the
    reference (frame.cpp:296-314), returns the same bits as the ctypes path."""
    import subprocess
    from test_abi import _build_adaptor
    H, W, nf = 120, 160, 200
    blob = weights.synthetic(7, "dense")
    img = synth.make_image(12, H, W)
    wpath, ipath, opath = tmp_path / "w.spfw", tmp_path / "im.raw", tmp_path / "out.bin"
    weights.save(wpath, blob)
    img.tofile(ipath)
    exe = _build_adaptor(tmp_path)
    subprocess.check_call([exe, str(wpath), str(ipath), str(H), str(W), str(nf), str(opath)])
    raw = np.fromfile(opath, np.uint8)
    K = int(raw[:4].view(np.int32)[0])
    off = 4
    kp = raw[off:off + K * 20].view(np.float32).reshape(K, 5); off += K * 20
    desc = raw[off:off + K * 1024].view(np.float32).reshape(K, 256); off += K * 1024
    C = (H // 8) * (W // 8)
    occ = raw[off:off + C * 2].view(np.int16).reshape(H // 8, W // 8); off += C * 2
    dust = raw[off:off + C * 4].view(np.float32).reshape(H // 8, W // 8); off += C * 4
    heat = raw[off:off + H * W * 4].view(np.float32).reshape(H, W)
    ext = SPExtractor(nf, H, W, blob)
    ext(img, None)
    fr = ext.last
    assert K == fr.K and np.array_equal(kp[:, :2], fr.kp_xy) and np.array_equal(kp[:, 2], fr.response)
    assert np.array_equal(kp[:, 3:], fr.cov2_inv) and np.array_equal(desc, fr.descriptors)
    assert np.array_equal(occ, fr.occ_grid) and np.array_equal(dust, fr.dense_dust)
    assert np.array_equal(heat, fr.heat)
    ext.close()


def test_cpp_dropin_through_base_pointer_matches_python_path(tmp_path):
    """orbslam::SPExtractor : BaseExtractor, spfe::ExtractorCV (include/orbslam_sp_extractor.hpp), constructed
    with the reference's 1-argument constructor and driven through a BaseExtractor* + dynamic_cast exactly as
    tracker.cpp:131 / frame.cpp:296-311 do (tests/cpp/dropin_main.cpp checks GetLevels() == 1, the scale
    vectors == {1}, getCov2Inv(), occ_grid_, the empty-image exception): same bits as the ctypes path."""
    import os
    import subprocess
    from test_abi import REF_CV_DIR, _build_dropin
    H, W, nf = 128, 160, 150
    blob = weights.synthetic(7, "dense")
    img = synth.make_image(13, H, W)
    wpath, ipath, opath = tmp_path / "w.spfw", tmp_path / "im.raw", tmp_path / "out.bin"
    weights.save(wpath, blob)
    img.tofile(ipath)
    exe = _build_dropin(tmp_path, os.path.exists(os.path.join(REF_CV_DIR, "base_extractor.h")))
    subprocess.check_call([exe, str(wpath), str(ipath), str(H), str(W), str(nf), str(opath)])
    raw = np.fromfile(opath, np.uint8)
    K = int(raw[:4].view(np.int32)[0])
    off = 4
    kp = raw[off:off + K * 20].view(np.float32).reshape(K, 5); off += K * 20
    desc = raw[off:off + K * 1024].view(np.float32).reshape(K, 256); off += K * 1024
    C = (H // 8) * (W // 8)
    occ = raw[off:off + C * 2].view(np.int16).reshape(H // 8, W // 8); off += C * 2
    dust = raw[off:off + C * 4].view(np.float32).reshape(H // 8, W // 8); off += C * 4
    heat = raw[off:off + H * W * 4].view(np.float32).reshape(H, W); off += H * W * 4
    heat_inv = raw[off:off + H * W * 4].view(np.float32).reshape(H, W)   # (heat_inv_ after operator(): filled by default; dropin_main also runs the lazy opt-in form)
    ref = oracle.extract(blob, img, nf)
    assert np.array_equal(heat_inv, ref["heat_inv"])
    assert K == ref["K"] and np.array_equal(kp[:, :2], ref["kp_xy"])
    assert np.array_equal(kp[:, 3:].view(np.uint32), ref["cov2_inv"].view(np.uint32))
    assert np.array_equal(desc.view(np.uint32), ref["desc"].view(np.uint32))
    assert np.array_equal(occ, ref["occ_grid"]) and np.array_equal(dust, ref["dense_dust"])
    assert np.array_equal(heat, ref["heat"]) and np.array_equal(kp[:, 2], ref["response"])


def test_lazy_heat_inv_is_fetched_on_demand():
    """SPFE_FLAG_LAZY_HEAT_INV: the host calls bring back `heat` only (spfe_result.heat_inv NULL) and spfe_fetch_heat_inv
    copies a frame's map on demand — the bits of the eager handle's; refused before the first call, for a frame the call did not
    hold, behind a call that was not a synchronous host call, and on a handle without SPFE_FLAG_HEAT."""
    H, W, nf, B = 120, 160, 100, 3
    blob = weights.synthetic(7, "dense")
    imgs = [synth.make_image(60 + i, H, W) for i in range(B)]
    eager = SPExtractor(nf, H, W, blob, max_batch=B)
    lazy = SPExtractor(nf, H, W, blob, max_batch=B, lazy_heat_inv=True)
    with pytest.raises(SpfeError):
        lazy.fetch_heat_inv(0)
    fe, fl = eager.extract_batch(imgs), lazy.extract_batch(imgs)
    for i in range(B):
        assert fl[i].heat_inv is None and np.array_equal(fl[i].heat, fe[i].heat)
        assert np.array_equal(lazy.fetch_heat_inv(i).view(np.uint32), fe[i].heat_inv.view(np.uint32))
        assert np.array_equal(fl[i].cov2, fe[i].cov2) and np.array_equal(fl[i].response, fe[i].response)
    lazy(imgs[1], None)
    assert np.array_equal(lazy.fetch_heat_inv(0), fe[1].heat_inv)
    with pytest.raises(SpfeError):
        lazy.fetch_heat_inv(1)       # the last call held one frame
    assert np.array_equal(eager.fetch_heat_inv(2), fe[2].heat_inv)   # works on an eager handle too
    # a call that is not a synchronous host call overwrites the device map (a device-resident call, a pipelined submission):
    # a fetch behind it is refused, not served from another call's map (enqueue() clears the host-call mark)
    import torch
    lazy.extract_batch(imgs)
    d = torch.from_numpy(np.stack(imgs)).cuda()
    lazy.extract_batch_device(d.data_ptr(), B)
    torch.cuda.synchronize()
    with pytest.raises(SpfeError):
        lazy.fetch_heat_inv(0)
    lazy.extract_batch(imgs)
    assert np.array_equal(lazy.fetch_heat_inv(2), fe[2].heat_inv)
    lazy.collect_batch(lazy.submit_batch(imgs))
    with pytest.raises(SpfeError):
        lazy.fetch_heat_inv(0)
    noheat = SPExtractor(nf, H, W, blob, with_heat=False)
    noheat(imgs[0], None)
    with pytest.raises(SpfeError):
        noheat.fetch_heat_inv(0)
    for e in (eager, lazy, noheat):
        e.close()


def test_python_record_layout_matches_library():
    from sp_orb_slam_amd import parallel
    for (H, W, nf) in [(64, 96, 30), (480, 752, 1000), (720, 1280, 800)]:
        ext = SPExtractor(nf, H, W, weights.synthetic(7, "dense"), with_heat=False)
        lay = parallel.RecordLayout(H, W, nf)
        for name in ("bytes", "kmax", "off_hdr", "off_xy", "off_resp", "off_cov", "off_cinv", "off_desc",
                     "off_occ", "off_dd", "off_sd"):
            assert getattr(lay, name) == getattr(ext.layout, name), name
        assert ext.layout.desc_elem_bytes == 4
        ext.close()
        ext = SPExtractor(nf, H, W, weights.synthetic(7, "dense"), with_heat=False, desc_bf16=True)
        lay = parallel.RecordLayout(H, W, nf, desc_bf16=True)
        for name in ("bytes", "kmax", "off_hdr", "off_xy", "off_resp", "off_cov", "off_cinv", "off_desc",
                     "off_occ", "off_dd", "off_sd"):
            assert getattr(lay, name) == getattr(ext.layout, name), name
        assert ext.layout.desc_elem_bytes == 2
        ext.close()


def test_device_path_sync_and_async_cov_equal_host_path():
    """spfe_extract_batch_device (records in device memory), with the covariance stage
    synchronous and overlapped (SPFE_FLAG_ASYNC_COV + spfe_wait_records): bit-identical
    to the host-facing call, batch after batch, through the pipelined ShardedExtractor."""
    import torch
    from sp_orb_slam_amd import parallel
    H, W, nf, B = 120, 160, 150, 3
    blob = weights.synthetic(7, "dense")
    batches = [np.stack([synth.make_image(60 + 10 * s + i, H, W) for i in range(B)]) for s in range(4)]
    host = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False)
    expect = [host.extract_batch(list(b)) for b in batches]
    host.close()
    stream = torch.cuda.current_stream()
    for async_cov in (False, True):
        ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, async_cov=async_cov)
        sh = parallel.ShardedExtractor(ext, 1, 0, B)
        d_batches = [torch.from_numpy(b).cuda() for b in batches]
        done = []
        for s, d in enumerate(d_batches):
            sh.step(d, stream)
            if async_cov:
                if s > 0:
                    torch.cuda.synchronize()
                    done.append([sh.decode(i) for i in range(B)])   # batch s-1 completed by step s
            else:
                torch.cuda.synchronize()
                done.append([sh.decode(i) for i in range(B)])
        if async_cov:
            sh.flush(stream)
            torch.cuda.synchronize()
            done.append([sh.decode(i) for i in range(B)])
        assert len(done) == len(batches)
        for got, exp in zip(done, expect):
            for g, e in zip(got, exp):
                assert g.status == 0 and g.K == e.K and np.array_equal(g.kp_xy, e.kp_xy)
                assert np.array_equal(g.descriptors, e.descriptors) and np.array_equal(g.occ_grid, e.occ_grid)
                assert np.array_equal(g.cov2, e.cov2) and np.array_equal(g.cov2_inv, e.cov2_inv)
                assert np.array_equal(g.response, e.response) and np.array_equal(g.dense_dust, e.dense_dust)
        ext.close()


def test_sharded_driver_comm_stream_and_buffer_rotation():
    """The N > 1 code path of ShardedExtractor (communication stream, rotating record buffers, two
    gather outputs) on one GPU: the collective is replaced by a device copy, everything else — the
    stream waits, spfe_wait_records on the communication stream, buffer reuse — is the real thing.
    Many steps back to back without host synchronisation, alternating inputs."""
    import torch
    from sp_orb_slam_amd import parallel
    H, W, nf, B = 120, 160, 150, 2
    blob = weights.synthetic(7, "dense")
    batches = [np.stack([synth.make_image(900 + 10 * s + i, H, W) for i in range(B)]) for s in range(3)]
    host = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False)
    expect = [host.extract_batch(list(b)) for b in batches]
    host.close()
    d_batches = [torch.from_numpy(b).cuda() for b in batches]
    torch.cuda.synchronize()
    calls = []

    def fake_gather(out, local):
        calls.append(torch.cuda.current_stream().cuda_stream)
        out.copy_(local)

    for async_cov in (False, True):
        ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, async_cov=async_cov)
        sh = parallel.ShardedExtractor(ext, 1, 0, B, gather_fn=fake_gather)
        calls.clear()
        comp, cons = torch.cuda.Stream(), torch.cuda.Stream()
        snaps, order = [], []
        nsteps = 11
        for k in range(nsteps):
            out = sh.step(d_batches[k % 3], comp)
            done = k - 1 if async_cov else k
            if done >= 0:
                sh.sync(cons)
                with torch.cuda.stream(cons):
                    snaps.append(out.clone())
                order.append(done)
        if async_cov:
            out = sh.flush(comp)
            sh.sync(cons)
            with torch.cuda.stream(cons):
                snaps.append(out.clone())
            order.append(nsteps - 1)
        torch.cuda.synchronize()
        assert order == list(range(nsteps))
        assert all(c == sh.comm.cuda_stream for c in calls) and sh.comm.cuda_stream != comp.cuda_stream
        rb = ext.record_bytes()
        for k, snap in zip(order, snaps):
            hrec = snap.cpu().numpy()
            for i, e in enumerate(expect[k % 3]):
                g = ext.view_record(hrec[i * rb:(i + 1) * rb])
                assert g.status == 0 and g.K == e.K and np.array_equal(g.kp_xy, e.kp_xy)
                assert np.array_equal(g.descriptors, e.descriptors) and np.array_equal(g.cov2_inv, e.cov2_inv)
        assert sh.decode(0).K == expect[(nsteps - 1) % 3][0].K
        ext.close()


def test_bench_workload_b8_async_matches_oracle():
    """The exact workload bench.py times (BASELINE configs[1] sharded like configs[2]): 8 frames of
    752x480 per call, num_features 1000, device-resident, covariance overlapped with the next call's
    convolutions, through ShardedExtractor — every record of a pipelined batch against the oracle:
    keypoints / occ_grid exact, descriptors / cov2 / response bitwise."""
    import torch
    from sp_orb_slam_amd import parallel
    H, W, nf, B = 480, 752, 1000, 8
    blob = weights.synthetic(7, "dense")
    frames = synth.make_batch(200, B, H, W)
    ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, async_cov=True)
    sh = parallel.ShardedExtractor(ext, 1, 0, B)
    d_img = torch.from_numpy(frames).cuda()
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):          # the last completed batch has two younger batches' kernels beside it
        sh.step(d_img, stream)
    sh.flush(stream)
    recs = [sh.decode(i) for i in range(B)]   # no global synchronize: decode() orders itself
    ext.close()
    for i in range(B):
        ref = oracle.extract(blob, frames[i], nf)
        g = recs[i]
        assert g.status == 0 and g.K == ref["K"] and g.n_candidates == ref["n_candidates"]
        assert np.array_equal(g.kp_xy, ref["kp_xy"]) and np.array_equal(g.occ_grid, ref["occ_grid"])
        assert np.array_equal(g.descriptors.view(np.uint32), ref["desc"].view(np.uint32))
        assert np.array_equal(g.cov2.view(np.uint32), ref["cov2"].view(np.uint32))
        assert np.array_equal(g.cov2_inv.view(np.uint32), ref["cov2_inv"].view(np.uint32))
        assert np.array_equal(g.response.view(np.uint32), ref["response"].view(np.uint32))


def test_sharded_decode_orders_itself_after_compute_stream():
    """world == 1, no gather: decode() right after step() on a non-default stream, with NO
    torch.cuda.synchronize() in between (the documented use), returns the finished records."""
    import torch
    from sp_orb_slam_amd import parallel
    H, W, nf, B = 240, 320, 300, 4
    blob = weights.synthetic(7, "dense")
    frames = synth.make_batch(40, B, H, W)
    host = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False)
    expect = host.extract_batch(list(frames))
    host.close()
    for async_cov in (False, True):
        ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, async_cov=async_cov)
        sh = parallel.ShardedExtractor(ext, 1, 0, B)
        d_img = torch.from_numpy(frames).cuda()
        stream = torch.cuda.Stream()
        torch.cuda.synchronize()
        for rep in range(5):
            sh.step(d_img, stream)
            if async_cov:
                sh.flush(stream)
            for i in (B - 1, 0):
                g, e = sh.decode(i), expect[i]
                assert g.status == 0 and g.K == e.K and np.array_equal(g.kp_xy, e.kp_xy)
                assert np.array_equal(g.descriptors, e.descriptors) and np.array_equal(g.cov2_inv, e.cov2_inv)
        ext.close()


def test_rccl_native_allgather_single_rank():
    """spfe_comm_init / spfe_allgather_records (ncclAllGather inside libspfe.so, librccl dlopen'ed) with a
    1-rank communicator: the RCCL code path of the multi-GPU batch mode — unique id, ncclCommInitRank on the
    handle's device, the library's communication stream waiting for the batch's ticket, the gather, the
    completion event — executed on the one GPU a test box has.  Pipelined driver, many steps, bit-identical
    to the host-facing call.  (N > 1 needs N GPUs: RCCL refuses two ranks on one device.)"""
    import torch
    from sp_orb_slam_amd import parallel
    H, W, nf, B = 120, 160, 150, 3
    blob = weights.synthetic(7, "dense")
    batches = [np.stack([synth.make_image(500 + 10 * s + i, H, W) for i in range(B)]) for s in range(3)]
    host = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False)
    expect = [host.extract_batch(list(b)) for b in batches]
    host.close()
    d_batches = [torch.from_numpy(b).cuda() for b in batches]
    for async_cov in (False, True):
        ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, async_cov=async_cov)
        assert ext.comm_stream() == 0
        sh = parallel.ShardedExtractor(ext, 1, 0, B, native_comm=True)
        assert ext.comm_stream() != 0 and sh.comm.cuda_stream == ext.comm_stream()
        comp = torch.cuda.Stream()
        torch.cuda.synchronize()
        done = []
        for k in range(7):
            sh.step(d_batches[k % 3], comp)
            kk = k - 1 if async_cov else k
            if kk >= 0:
                done.append((kk, [sh.decode(i) for i in range(B)]))   # decode() waits for the gather's event
        if async_cov:
            sh.flush(comp)
            done.append((6, [sh.decode(i) for i in range(B)]))
        assert [k for k, _ in done] == list(range(7))
        for k, got in done:
            for g, e in zip(got, expect[k % 3]):
                assert g.status == 0 and g.K == e.K and np.array_equal(g.kp_xy, e.kp_xy)
                assert np.array_equal(g.descriptors, e.descriptors) and np.array_equal(g.cov2_inv, e.cov2_inv)
                assert np.array_equal(g.occ_grid, e.occ_grid)
        # what RCCL itself says about the communicator, and the collective timed alone (bench.py's N > 1 fields)
        assert ext.comm_count() == 1 and sh.comm_ranks()["library"] == 1
        tg = sh.time_gather(5)
        assert tg["ms"] > 0 and tg["bytes_per_rank"] == B * ext.record_bytes() and tg["iters"] == 5
        again = [sh.decode(i) for i in range(B)]           # the timed gathers re-gathered the last batch: same records
        for g, e in zip(again, expect[6 % 3]):
            assert g.K == e.K and np.array_equal(g.descriptors, e.descriptors)
        # errors: double init, bad ticket
        with pytest.raises(SpfeError):
            ext.comm_init(ext.comm_unique_id(), 0, 1)
        with pytest.raises(SpfeError):
            ext.allgather_records(10 ** 6, sh.local[0].data_ptr(), sh.all[0].data_ptr(), B)
        ext.comm_destroy()
        assert ext.comm_stream() == 0
        ext.close()


def test_pipelined_host_path_submit_collect():
    """spfe_submit_batch / spfe_collect_batch: three batches in flight, out-of-order collection, strided
    inputs, heat maps on and off — bit-identical to the synchronous host call; a fourth submit without a
    collect is refused."""
    H, W, nf, B = 120, 160, 150, 3
    blob = weights.synthetic(7, "dense")
    batches = [[synth.make_image(700 + 10 * s + i, H, W) for i in range(B)] for s in range(5)]
    for with_heat in (False, True):
        host = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=with_heat)
        expect = [host.extract_batch(b) for b in batches]
        host.close()
        ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=with_heat)
        wide = np.zeros((H, W + 32), np.uint8)
        tickets = []
        for s in range(3):
            imgs = batches[s]
            if s == 1:   # strided rows (cv::Mat::step > cols)
                imgs = []
                for im in batches[s]:
                    w2 = wide.copy()
                    w2[:, :W] = im
                    imgs.append(w2[:, :W])
            tickets.append(ext.submit_batch(imgs))
        with pytest.raises(SpfeError):
            ext.submit_batch(batches[3])       # pipeline full
        got = {1: ext.collect_batch(tickets[1])}              # out of order
        with pytest.raises(SpfeError):
            ext.submit_batch(batches[3])       # the ring's next slot is the oldest batch's: still in flight
        got[0] = ext.collect_batch(tickets[0])
        tickets.append(ext.submit_batch(batches[3]))
        tickets.append(ext.submit_batch(batches[4][:2]))   # a short batch
        got[2] = ext.collect_batch(tickets[2])
        got[3] = ext.collect_batch(tickets[3])
        got[4] = ext.collect_batch(tickets[4])
        with pytest.raises((SpfeError, KeyError)):
            ext.collect_batch(tickets[4])
        for s in range(5):
            exp = expect[s][:len(got[s])]
            assert len(got[s]) == (2 if s == 4 else B)
            for g, e in zip(got[s], exp):
                assert g.status == 0 and g.K == e.K and np.array_equal(g.kp_xy, e.kp_xy)
                assert np.array_equal(g.descriptors, e.descriptors) and np.array_equal(g.cov2_inv, e.cov2_inv)
                assert np.array_equal(g.occ_grid, e.occ_grid) and np.array_equal(g.response, e.response)
                if with_heat:
                    assert np.array_equal(g.heat, e.heat) and np.array_equal(g.heat_inv, e.heat_inv)
        ext.close()


def test_f32_heads_with_register_resident_weights_are_bit_identical(monkeypatch):
    """head_f32.hip (SPFE_F32_HEADS=1: convPb / convDb with the weights in registers) against the generic 1x1 path: the
    same k-ordered fmaf chain, so semi and coarse — and with them everything downstream — are the same bits."""
    H, W, nf = 240, 376, 500
    blob = weights.synthetic(7, "dense")
    imgs = [synth.make_image(60 + i, H, W) for i in range(3)]
    out = {}
    monkeypatch.setenv("SPFE_SPARSE_DB", "0")   # (the gathered descriptor head IS head_f32.hip's kernel: compare the dense launches)
    for flag in ("0", "1"):
        monkeypatch.setenv("SPFE_F32_HEADS", flag)
        ext = SPExtractor(nf, H, W, blob, max_batch=3, with_heat=False)
        frs = ext.extract_batch(imgs)
        out[flag] = (frs, [ext.debug_read("semi", i) for i in range(3)], [ext.debug_read("coarse", i) for i in range(3)])
        ext.close()
    for a, b in zip(out["0"][1] + out["0"][2], out["1"][1] + out["1"][2]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for a, b in zip(out["0"][0], out["1"][0]):
        assert np.array_equal(a.kp_xy, b.kp_xy) and np.array_equal(a.descriptors, b.descriptors)


@pytest.mark.parametrize("H,W,B", [(240, 376, 1), (120, 168, 3), (480, 752, 1)])
def test_f32_two_row_tiles_are_bit_identical(monkeypatch, H, W, B):
    """2-row tiles of conv_f32_kernel (chosen by the cost model for single frames; forced here on every layer without a
    pool, SPFE_TILE2_MASK=0xEA) against the 4 / 8-row tiles (SPFE_TILE2_MASK=0: never) and the cost model (unset): the tile shape does not touch an output's K order, so semi,
    coarse and the intermediate activations are the same bits — ragged heights (120 / 8 = 15 rows) included."""
    nf = 300
    blob = weights.synthetic(7, "dense")
    imgs = [synth.make_image(70 + i, H, W) for i in range(B)]
    out = {}
    for name, mask in (("tall", "0"), ("two", "0xEA"), ("auto", None)):
        if mask is None:
            monkeypatch.delenv("SPFE_TILE2_MASK", raising=False)
        else:
            monkeypatch.setenv("SPFE_TILE2_MASK", mask)
        ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False)
        frs = ext.extract_batch(imgs)
        out[name] = (frs, [ext.debug_read(nm, i) for i in range(B) for nm in ("semi", "coarse", "feat")])
        ext.close()
    for name in ("two", "auto"):
        for a, b in zip(out["tall"][1], out[name][1]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        for a, b in zip(out["tall"][0], out[name][0]):
            assert np.array_equal(a.kp_xy, b.kp_xy) and np.array_equal(a.descriptors, b.descriptors)
            assert np.array_equal(a.cov2, b.cov2)


@pytest.mark.parametrize("H,W,B", [(240, 376, 2), (120, 168, 3), (136, 200, 1), (480, 752, 1), (24, 40, 2)])
def test_f32_conv1b_16_row_tiles_are_bit_identical(monkeypatch, H, W, B):
    """conv1b on 16-row tiles (4 wavefronts x 4 rows x 64 channels: the cost model's choice for the large launches; forced
    here) against the 8-row tiles: the pooled output act1 and everything behind it are the same bits, ragged heights
    (120 = 7.5 tiles, 136 = 8.5, 24 = 1.5) included."""
    blob = weights.synthetic(7, "dense")
    imgs = [synth.make_image(90 + i, H, W) for i in range(B)]
    out = {}
    for flag in ("0", "2"):
        monkeypatch.setenv("SPFE_TILE16X4", flag)
        ext = SPExtractor(200, H, W, blob, max_batch=B, with_heat=False)
        frs = ext.extract_batch(imgs)
        out[flag] = (frs, [ext.debug_read(nm, i) for i in range(B) for nm in ("act1", "semi", "feat")])
        ext.close()
    for a, b in zip(out["0"][1], out["2"][1]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for a, b in zip(out["0"][0], out["2"][0]):
        assert np.array_equal(a.kp_xy, b.kp_xy) and np.array_equal(a.descriptors, b.descriptors)


@pytest.mark.parametrize("H,W,B,rows", [(480, 752, 8, 224), (488, 752, 8, None), (720, 1280, 4, None)])
def test_f32_conv1b_cut_in_a_16_row_and_an_8_row_launch_is_bit_identical(monkeypatch, H, W, B, rows):
    """Large launches whose work list divides neither way: the first k tile rows of the batch as 16-row tiles, the rest as
    8-row tiles in a second launch (752x480 x 8: k = 224 of 240).  Same bits as 8-row tiles alone — the frame the cut goes
    through and a height that is not a multiple of 16 included."""
    blob = weights.synthetic(7, "dense")
    imgs = [synth.make_image(130 + i, H, W) for i in range(B)]
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("SPFE_TILE16X4", flag)
        ext = SPExtractor(100, H, W, blob, max_batch=B, with_heat=False)
        ext.extract_batch(imgs)
        out[flag] = [ext.debug_read("act1", i) for i in range(B)] + [ext.debug_read("semi", B - 1)]
        if flag == "1":
            k = int(ext.debug_read("conv1b_split_rows")[0])
            if rows is not None:
                assert k == rows
            out["k"] = k
        ext.close()
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_debug_read_follows_the_double_buffered_tail_outputs():
    """heat_log / cell_score exist twice (by ticket parity, so that the next batch's detector tail does not wait for this
    batch's side chain): debug_read must hand out the set the LAST call wrote, whichever parity that was."""
    H, W = 120, 160
    blob = weights.synthetic(7, "dense")
    a, b = synth.make_image(81, H, W), synth.make_image(82, H, W)
    one = SPExtractor(100, H, W, blob, with_heat=False)
    one(b, None)                                    # ticket 0
    ref = {nm: one.debug_read(nm) for nm in ("heat_log", "cell_score", "semi")}
    one.close()
    two = SPExtractor(100, H, W, blob, with_heat=False)
    two(a, None)                                    # ticket 0
    two(b, None)                                    # ticket 1: the other set
    for nm, r in ref.items():
        assert np.array_equal(two.debug_read(nm).view(np.uint32), r.view(np.uint32)), nm
    two(a, None)                                    # ticket 2: back to the first set, now holding image a
    assert not np.array_equal(two.debug_read("heat_log"), ref["heat_log"])
    two(b, None)
    assert np.array_equal(two.debug_read("heat_log").view(np.uint32), ref["heat_log"].view(np.uint32))
    two.close()


@pytest.mark.parametrize("H,W,B", [(480, 752, 1), (240, 376, 3), (136, 200, 2), (24, 40, 2), (64, 96, 1)])
def test_f32_convPb_inside_the_tail_launch_is_bit_identical(monkeypatch, H, W, B):
    """pbtail_f32.hip (round 4): convPb's two full channel tiles on the MFMA, its dustbin channel as an fmaf chain on the VALU,
    the detector tail on the logits while they sit in LDS — against convPb as a launch of the generic kernel + tail_kernel
    (SPFE_PBTAIL=0).  Same logits (all 65 channels), same heat maps, dust maps, scores and records; frames whose cell count
    is not a multiple of the 32-cell tile (5640, 1410, 425, 15, 96) and batches included.  Both against the oracle."""
    nf = 300
    blob = weights.synthetic(7, "dense")
    imgs = [synth.make_image(170 + i, H, W) for i in range(B)]
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("SPFE_PBTAIL", flag)
        ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=True)
        frs = ext.extract_batch(imgs)
        out[flag] = (frs, [ext.debug_read(nm, i) for i in range(B) for nm in ("semi", "heat_log", "cell_score")])
        ext.close()
    for a, b in zip(out["0"][1], out["1"][1]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for a, b in zip(out["0"][0], out["1"][0]):
        assert a.K == b.K and np.array_equal(a.kp_xy, b.kp_xy) and np.array_equal(a.descriptors, b.descriptors)
        assert np.array_equal(a.cov2, b.cov2) and np.array_equal(a.occ_grid, b.occ_grid)
        assert np.array_equal(a.dense_dust, b.dense_dust) and np.array_equal(a.semi_dust, b.semi_dust)
        assert np.array_equal(a.heat, b.heat) and np.array_equal(a.heat_inv, b.heat_inv)
    ref = oracle.extract(blob, imgs[B - 1], nf)
    last = out["1"][0][B - 1]
    assert last.K == ref["K"] and np.array_equal(last.kp_xy, ref["kp_xy"])
    assert np.array_equal(last.descriptors.view(np.uint32), ref["desc"].view(np.uint32))
    assert np.array_equal(last.semi_dust.view(np.uint32), ref["semi_dust"].view(np.uint32))
    assert np.array_equal(last.heat.view(np.uint32), ref["heat"].view(np.uint32))


@pytest.mark.parametrize("H,W", [(480, 752), (240, 368), (64, 96)])
def test_f32_pooled_layer_as_unpooled_two_row_tiles_plus_a_pool_pass_is_bit_identical(monkeypatch, H, W):
    """Single frames (round 4): conv2b / conv3b as UN-pooled 2-row tiles into a scratch buffer + pool2x2_f32_kernel
    (SPFE_POOL_SPLIT=1: wherever the shapes allow; the cost model takes it for conv3b at 752x480) against the fused
    bias / ReLU / 2x2-max epilogue (SPFE_POOL_SPLIT=0): the pooled activations and everything behind them are the same bits."""
    blob = weights.synthetic(7, "dense")
    img = synth.make_image(190, H, W)
    out = {}
    for flag in ("0", "1", "-1"):
        monkeypatch.setenv("SPFE_POOL_SPLIT", flag)
        ext = SPExtractor(300, H, W, blob, max_batch=1, with_heat=False)
        fr = ext(img, None)
        out[flag] = (ext.last, [ext.debug_read(nm, 0) for nm in ("act3", "act5", "semi", "feat")])
        ext.close()
    for flag in ("1", "-1"):
        for a, b in zip(out["0"][1], out[flag][1]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        a, b = out["0"][0], out[flag][0]
        assert a.K == b.K and np.array_equal(a.kp_xy, b.kp_xy) and np.array_equal(a.descriptors, b.descriptors)
        assert np.array_equal(a.cov2, b.cov2)


@pytest.mark.parametrize("cfg", ["f32 1", "f32 4", "bf16 1", "bf16 3", "bf16 2 720 1280"])
def test_a_synchronous_call_captured_into_a_hip_graph_replays_bit_identically(cfg):
    """tools/graph_capture_check.py (its own process: a failed capture leaves the runtime in capture mode): one
    spfe_extract_batch_device call captured on a torch stream — the library's side streams and its second convolution stream
    fork from and join the capturing stream by events; the selection's extended-launch stop event (not a capture node) gives
    way to a plain record; waits are never replaced by host-side event queries there — and replayed on three different frame
    sets: every field of every record equals the direct call's."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "graph_capture_check.py")] + cfg.split(), cwd=root,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "bit-identical to direct calls: True" in out.stdout


@pytest.mark.parametrize("cfg", ["f32 2", "bf16 2"])
def test_direct_calls_behind_graph_replays_across_a_generation_wrap(cfg):
    """tools/graph_capture_check.py wrap — the claim / done maps' generation codes start at 5
    (SPFE_COV_CAPS field 6), a call is captured at code 2, direct calls run across the wrap, and direct calls follow replays
    with codes above, at and below the captured one: every record equals a fresh default handle's (the stale, lower G-tagged
    entries a replay leaves behind must not win a later direct call's claims)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SPFE_COV_CAPS=",,,,,5")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "graph_capture_check.py"), "wrap"] + cfg.split(), cwd=root,
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "every call bit-identical to a fresh handle's: True" in out.stdout


def test_heat_maps_sent_ahead_of_the_record_are_the_same_maps(monkeypatch):
    """Round 6: a synchronous host call with heat maps (the drop-in's operator(): Frame clones heat_, frame.cpp:304) starts the
    maps' D2H behind the heat normalisation's completion signal, on a copy stream of its own, beside selection + covariance
    instead of behind the record (SPFE_EARLY_HEAT_COPY=0: as before).  Same maps, same records, call after call on changing
    frames — single frames, batches, the lazy heat_inv form and spfe_postprocess."""
    H, W, nf, B = 240, 376, 300, 3
    blob = weights.synthetic(7, "dense")
    frames = [[synth.make_image(2100 + 7 * r + i, H, W) for i in range(B)] for r in range(4)]
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("SPFE_EARLY_HEAT_COPY", flag)
        ext = SPExtractor(nf, H, W, blob, max_batch=B)
        lazy = SPExtractor(nf, H, W, blob, max_batch=B, lazy_heat_inv=True)
        res = []
        for r in range(4):
            for fr in ext.extract_batch(frames[r]):
                res.append((fr.K, fr.kp_xy.copy(), fr.descriptors.copy(), fr.cov2.copy(), fr.heat.copy(), fr.heat_inv.copy()))
            ext(frames[r][0], None)
            res.append((ext.last.K, ext.last.kp_xy.copy(), ext.last.descriptors.copy(), ext.last.cov2.copy(), ext.heat_.copy(), ext.heat_inv_.copy()))
            fl = lazy.extract_batch(frames[r])
            assert all(x.heat_inv is None for x in fl)
            res.append((fl[1].K, fl[1].kp_xy.copy(), fl[1].descriptors.copy(), fl[1].cov2.copy(), fl[1].heat.copy(), lazy.fetch_heat_inv(1).copy()))
        ext.close()
        lazy.close()
        out[flag] = res
    monkeypatch.delenv("SPFE_EARLY_HEAT_COPY")
    assert len(out["0"]) == len(out["1"]) == 4 * (B + 2)
    for a, b in zip(out["0"], out["1"]):
        assert a[0] == b[0] and a[0] > 0
        for x, y in zip(a[1:], b[1:]):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    ref = oracle.extract(blob, frames[3][0], nf)       # ... and they are the oracle's maps
    assert np.array_equal(out["1"][-2][4], ref["heat"]) and np.array_equal(out["1"][-2][5], ref["heat_inv"])


def test_the_call_in_three_parts_is_the_call(monkeypatch):
    """spfe_extract_begin / spfe_extract_maps / spfe_extract_finish (spfe.h; what the drop-in's operator() runs so that its deep
    copies of heat_ / heat_inv_ go beside the device's selection + covariance): begin + finish is spfe_extract_batch bit for bit,
    the maps handed out early are the maps of the result, with SPFE_EARLY_HEAT_COPY=0 / without SPFE_FLAG_HEAT none are handed
    out early, and the handle refuses anything else while a call is open."""
    H, W, nf, B = 240, 376, 300, 3
    blob = weights.synthetic(7, "dense")
    frames = [[synth.make_image(2300 + 5 * r + i, H, W) for i in range(B)] for r in range(3)]
    same = lambda x, y: np.array_equal(np.ascontiguousarray(x).view(np.uint32), np.ascontiguousarray(y).view(np.uint32))
    for flag, heat, lazy in (("1", True, False), ("1", True, True), ("0", True, False), ("1", False, False)):
        monkeypatch.setenv("SPFE_EARLY_HEAT_COPY", flag)
        ref = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=heat, lazy_heat_inv=lazy)
        ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=heat, lazy_heat_inv=lazy)
        for r in range(3):
            batch = frames[r][: (B, 1, 2)[r]]
            want = ref.extract_batch(batch)
            ext.extract_begin(batch)
            with pytest.raises(Exception, match="open"):
                ext.extract_begin(batch)
            with pytest.raises(Exception, match="open"):
                ext.extract_batch(batch)
            hm, hi = ext.extract_maps()
            rows = [ext.extract_rows(i) for i in range(len(batch))]
            assert flag == "1" or all(x is None for x in rows)
            assert not (flag == "1" and len(batch) == 1) or rows[0] is not None   # (the drop-in's call: a single frame, the inline chain)
            rows = [x.copy() if x is not None else None for x in rows]
            with pytest.raises(Exception, match="not one of"):
                ext.extract_rows(len(batch))
            early = heat and flag == "1"
            assert (hm is not None) == early and (hi is not None) == (early and not lazy)
            hm = hm.copy() if early else None
            hi = hi.copy() if hi is not None else None
            got = ext.extract_finish()
            assert len(got) == len(want)
            for i, (a, b) in enumerate(zip(got, want)):
                assert a.K == b.K and a.K > 0
                for f in ("kp_xy", "response", "descriptors", "cov2", "cov2_inv", "dense_dust", "semi_dust"):
                    assert same(getattr(a, f), getattr(b, f)), f
                assert np.array_equal(a.occ_grid, b.occ_grid)
                if rows[i] is not None:
                    assert rows[i].shape == (b.K, 256) and same(rows[i], b.descriptors)
                if heat:
                    assert same(a.heat, b.heat)
                    if early:
                        assert same(hm[i], b.heat)
                    if not lazy:
                        assert same(a.heat_inv, b.heat_inv)
                        if early:
                            assert same(hi[i], b.heat_inv)
            with pytest.raises(Exception, match="no open call"):
                ext.extract_maps()
            with pytest.raises(Exception, match="no open call"):
                ext.extract_finish()
        ext.close()
        ref.close()
    monkeypatch.delenv("SPFE_EARLY_HEAT_COPY")


def test_maps_straight_into_the_callers_memory():
    """spfe_set_map_buffers (what the drop-in's operator() uses: heat_ / heat_inv_ are cv::Mat members the reference re-fills per
    call, sp_extractor.cpp:461-474): the maps of the synchronous calls land in the caller's arrays — page-locked by the library,
    not page-aligned here — bit for bit the maps of a handle without them; batches, single frames, the three-part call, the lazy
    form's fetch, other arrays, and the library's own buffers again."""
    H, W, nf, B = 240, 376, 300, 3
    blob = weights.synthetic(7, "dense")
    frames = [[synth.make_image(2500 + 5 * r + i, H, W) for i in range(B)] for r in range(3)]
    same = lambda x, y: np.array_equal(np.ascontiguousarray(x).view(np.uint32), np.ascontiguousarray(y).view(np.uint32))
    ref = SPExtractor(nf, H, W, blob, max_batch=B)
    ext = SPExtractor(nf, H, W, blob, max_batch=B)
    backing = [np.full(B * H * W + 3, np.nan, np.float32) for _ in range(4)]
    bufs = [b[3:].reshape(B, H, W) for b in backing]          # 12 bytes off whatever alignment numpy gave
    ext.set_map_buffers(bufs[0], bufs[1])
    for r in range(3):
        want = ref.extract_batch(frames[r])
        got = ext.extract_batch(frames[r])
        for i in range(B):
            assert same(bufs[0][i], want[i].heat) and same(bufs[1][i], want[i].heat_inv)
            assert same(got[i].heat, want[i].heat) and same(got[i].heat_inv, want[i].heat_inv)
            assert got[i].K == want[i].K and same(got[i].descriptors, want[i].descriptors) and same(got[i].cov2, want[i].cov2)
    ext.set_map_buffers(bufs[2], bufs[3])                     # other arrays: the first pair is left alone from here on
    keep = bufs[0].copy()
    ext.extract_begin(frames[0][:1])
    with pytest.raises(Exception, match="open"):
        ext.set_map_buffers(None, None)
    hm, hi = ext.extract_maps()
    assert hm is not None and hm.ctypes.data == bufs[2].ctypes.data and hi.ctypes.data == bufs[3].ctypes.data
    got = ext.extract_finish()
    want = ref.extract_batch(frames[0][:1])
    assert same(bufs[2][0], want[0].heat) and same(bufs[3][0], want[0].heat_inv) and same(got[0].heat_inv, want[0].heat_inv)
    assert same(bufs[0], keep)
    ext.set_map_buffers(bufs[2], None)                        # one map only; then neither
    got = ext.extract_batch(frames[1])
    want = ref.extract_batch(frames[1])
    assert same(bufs[2][2], want[2].heat) and same(got[2].heat_inv, want[2].heat_inv)
    ext.set_map_buffers(None, None)
    snap = bufs[2].copy()
    got = ext.extract_batch(frames[2])
    want = ref.extract_batch(frames[2])
    assert same(got[1].heat, want[1].heat) and same(got[1].heat_inv, want[1].heat_inv) and same(bufs[2], snap)
    ext.close()
    lazy = SPExtractor(nf, H, W, blob, max_batch=B, lazy_heat_inv=True)
    lazy.set_map_buffers(bufs[0], bufs[1])
    got = lazy.extract_batch(frames[2])
    assert all(g.heat_inv is None for g in got) and same(bufs[0][1], want[1].heat)
    assert same(lazy.fetch_heat_inv(1), want[1].heat_inv) and same(bufs[1][1], want[1].heat_inv)
    lazy.close()
    nomaps = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False)
    with pytest.raises(Exception, match="SPFE_FLAG_HEAT"):
        nomaps.set_map_buffers(bufs[0], None)
    nomaps.close()
    ref.close()
