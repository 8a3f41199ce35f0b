"""GPU: direct "dust" alignment (spfe_align_dust, csrc/dust.hip) against the CPU oracle
(oracle_align_dust — Optimizer::PoseOptimizationDust, optimizer_dust.cpp:170-294; g2o restated, PARITY UNPINNED).

Both sides evaluate include/spfe_dust_math.h and sum the edges by the same fixed-shape tree (round 5), so they differ only
through the device's sin / cos / sqrt / division in the exponential map and the solve (<= 1 ulp each).  Tolerances:
pose 1e-5 (absolute, 4x4 float), projections 1e-3 cell, inlier flags equal except where chi2 is within 1e-6 of
the 0.9 threshold, iteration count equal."""
import numpy as np
import pytest

from oracle import oracle
from sp_orb_slam_amd import synth, weights
from tools import dust_scene
from sp_orb_slam_amd.extractor import DUST_OUT_BYTES, SPExtractor, SpfeError

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-5
UV_TOL = 1e-3


def _compare(g, r, dust):
    assert g["iterations"] == r["iterations"]
    assert np.abs(g["Tcw"] - r["Tcw"]).max() <= POSE_TOL
    both = g["inlier"] & r["inlier"]
    assert np.abs(g["uv"][both] - r["uv"][both]).max(initial=0) <= UV_TOL
    diff = np.flatnonzero(g["inlier"] != r["inlier"])
    for i in diff:   # only threshold-straddling edges may flip
        u, v = r["uv"][i]
        xf, yf = int(np.floor(u)), int(np.floor(v))
        xx, yy = u - xf, v - yf
        val = ((1 - xx) * (1 - yy) * dust[yf, xf] + xx * (1 - yy) * dust[yf, xf + 1] + (1 - xx) * yy * dust[yf + 1, xf] +
               xx * yy * dust[yf + 1, xf + 1])
        assert abs(val * val - 0.9) < 1e-4, i
    assert abs(g["n_inlier"] - r["n_inlier"]) <= len(diff)
    assert g["n_inlier"] == int(g["inlier"].sum())


@pytest.mark.parametrize("H,W,n,seed", [(480, 752, 160, 0), (480, 752, 200, 1), (480, 640, 97, 2), (720, 1280, 300, 3),
                                        (480, 752, 512, 4), (480, 752, 1, 5), (120, 160, 40, 6),
                                        (1080, 1920, 160, 8),     # 32,400 cells: the largest standard map that fits in LDS
                                        (1440, 2560, 160, 9)])    # 57,600 cells: the map does not fit in LDS, read through L2
def test_align_dust_matches_oracle(H, W, n, seed):
    sc = dust_scene.make_scene(seed, H=H, W=W, n_points=n, cx=W / 2 - 8.8, cy=H / 2 + 8.4)
    ext = SPExtractor(100, H, W, weights.synthetic(7, "dense"), with_heat=False)
    g = ext.align_dust(sc["dust"], sc["pts"], sc["Tcw_init"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
    r = oracle.align_dust(sc["dust"], sc["pts"], sc["Tcw_init"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
    _compare(g, r, sc["dust"])
    # the alignment did something: the projections moved towards the keypoints
    uv_t, _ = dust_scene.project(sc["Tcw_true"], sc["pts"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
    uv_0, _ = dust_scene.project(sc["Tcw_init"], sc["pts"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
    uv_1, _ = dust_scene.project(g["Tcw"], sc["pts"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
    if n >= 40:
        assert np.abs(uv_1 - uv_t).mean() < np.abs(uv_0 - uv_t).mean()
    # other iteration caps / kernel parameters
    for it, delta in ((0, 0.9), (3, 0.9), (40, 0.3)):
        g2 = ext.align_dust(sc["dust"], sc["pts"], sc["Tcw_init"], sc["fx"], sc["fy"], sc["cx"], sc["cy"],
                            max_iterations=it, huber_delta=delta)
        r2 = oracle.align_dust(sc["dust"], sc["pts"], sc["Tcw_init"], sc["fx"], sc["fy"], sc["cx"], sc["cy"],
                               max_iterations=it, delta=delta)
        _compare(g2, r2, sc["dust"])
    ext.close()


def test_align_dust_edge_cases():
    H, W = 480, 752
    sc = dust_scene.make_scene(11, n_points=64, outlier_frac=0.0)
    ext = SPExtractor(100, H, W, weights.synthetic(7, "dense"), with_heat=False)
    # points behind the camera / outside the map, a flat map (H = 0: ten failed trials, Terminate)
    pts = sc["pts"].copy()
    T = sc["Tcw_init"].astype(np.float64)
    pts[0] = (np.array([0.1, -0.2, -3.0]) - T[:3, 3]) @ T[:3, :3]
    pts[1] = (np.array([-30.0, 0.0, 4.0]) - T[:3, 3]) @ T[:3, :3]
    for dust in (sc["dust"], np.full_like(sc["dust"], 0.5)):
        g = ext.align_dust(dust, pts, sc["Tcw_init"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
        r = oracle.align_dust(dust, pts, sc["Tcw_init"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
        _compare(g, r, dust)
        assert not g["inlier"][0] and not g["inlier"][1]
    # n = 0
    g = ext.align_dust(sc["dust"], np.zeros((0, 3), np.float32), sc["Tcw_init"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
    assert g["n_inlier"] == 0 and np.abs(g["Tcw"] - sc["Tcw_init"]).max() < 1e-6
    with pytest.raises(SpfeError):
        ext.align_dust(sc["dust"], np.zeros((513, 3), np.float32), sc["Tcw_init"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
    with pytest.raises(SpfeError):
        ext.align_dust(sc["dust"][:-1], sc["pts"], sc["Tcw_init"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
    ext.close()


def test_align_dust_on_a_resident_record():
    """The device form: the dust map is the dense_dust of a record that never left HBM (extraction -> alignment)."""
    import torch
    H, W, nf = 240, 320, 300
    blob = weights.synthetic(7, "sparse")
    img = synth.make_image(21, H, W)
    ext = SPExtractor(nf, H, W, blob, with_heat=False, async_cov=True)
    d_img = torch.from_numpy(img[None]).cuda()
    d_rec = torch.zeros(ext.record_bytes(), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.Stream()
    t = ext.extract_batch_device(d_img.data_ptr(), 1, d_rec.data_ptr(), stream.cuda_stream)
    ext.wait_records(t, stream.cuda_stream)
    sc = dust_scene.make_scene(5, H=H, W=W, n_points=120, fx=200.0, fy=200.0, cx=W / 2 - 3.0, cy=H / 2 + 2.0)
    d_pts = torch.from_numpy(sc["pts"]).cuda()
    d_T = torch.from_numpy(sc["Tcw_init"].reshape(16)).cuda()
    d_out = torch.zeros(DUST_OUT_BYTES, dtype=torch.uint8, device="cuda")
    ext.align_dust_record_device(d_rec.data_ptr(), d_pts.data_ptr(), 120, d_T.data_ptr(), d_out.data_ptr(), 200.0, 200.0,
                                 sc["cx"], sc["cy"], stream=stream.cuda_stream)
    stream.synchronize()
    rec = ext.view_record(d_rec.cpu().numpy())
    g = ext.decode_dust_out(d_out.cpu().numpy(), 120)
    r = oracle.align_dust(rec.dense_dust, sc["pts"], sc["Tcw_init"], 200.0, 200.0, sc["cx"], sc["cy"])
    _compare(g, r, rec.dense_dust)
    ext.close()


def test_align_dust_batch_of_records():
    """spfe_align_dust_batch_device: independent solves side by side, one workgroup per frame — each equals the oracle on its
    own record / points / start pose, including frames with 0 and 1 points."""
    import torch
    from sp_orb_slam_amd import parallel
    from sp_orb_slam_amd.extractor import DUST_MAX_POINTS
    H, W, nf, NB = 240, 320, 300, 6
    blob = weights.synthetic(7, "sparse")
    ext = SPExtractor(nf, H, W, blob, with_heat=False)
    lay = parallel.RecordLayout(H, W, nf)
    recs = np.zeros((NB, ext.record_bytes()), np.uint8)
    pts = np.zeros((NB, DUST_MAX_POINTS, 3), np.float32)
    T = np.zeros((NB, 16), np.float32)
    npts = np.array([120, 0, 1, 200, 64, 512], np.int32)
    scenes = []
    for f in range(NB):
        sc = dust_scene.make_scene(30 + f, H=H, W=W, n_points=max(int(npts[f]), 1), fx=200.0, fy=200.0, cx=W / 2 - 3.0, cy=H / 2 + 2.0)
        scenes.append(sc)
        recs[f, lay.off_dd:lay.off_dd + sc["dust"].size * 4] = sc["dust"].reshape(-1).view(np.uint8)
        pts[f, :npts[f]] = sc["pts"][:npts[f]]
        T[f] = sc["Tcw_init"].reshape(16)
    d_recs, d_pts, d_T = torch.from_numpy(recs).cuda(), torch.from_numpy(pts).cuda(), torch.from_numpy(T).cuda()
    d_n = torch.from_numpy(npts).cuda()
    d_out = torch.zeros((NB, DUST_OUT_BYTES), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.Stream()
    ext.align_dust_batch_device(d_recs.data_ptr(), NB, d_pts.data_ptr(), d_n.data_ptr(), d_T.data_ptr(), d_out.data_ptr(),
                                200.0, 200.0, scenes[0]["cx"], scenes[0]["cy"], stream=stream.cuda_stream)
    stream.synchronize()
    out = d_out.cpu().numpy()
    for f in range(NB):
        n = int(npts[f])
        g = ext.decode_dust_out(out[f], n)
        r = oracle.align_dust(scenes[f]["dust"], pts[f, :n], scenes[f]["Tcw_init"], 200.0, 200.0, scenes[f]["cx"], scenes[f]["cy"])
        if n == 0:
            assert g["n_inlier"] == 0 and np.abs(g["Tcw"] - scenes[f]["Tcw_init"]).max() < 1e-6
        else:
            _compare(g, r, scenes[f]["dust"])
    ext.close()
