"""CPU: the dust-alignment oracle (oracle_align_dust, a restatement of Optimizer::PoseOptimizationDust,
/root/reference/orb_slam2/src/mapping/optimizer_dust.cpp:170-294, and of the g2o pieces it drives — g2o is not in
the reference snapshot: PARITY UNPINNED) on known-answer cases and against an independent numpy statement of
the edge (types_dust_tracking.cpp:37-140)."""
import numpy as np

from oracle import oracle
from tools import dust_scene


def _bilinear(d, x, y):
    x, y = np.float32(x), np.float32(y)
    xf, yf = int(np.floor(x)), int(np.floor(y))
    xx, yy = np.float32(x - xf), np.float32(y - yf)
    one = np.float32(1)
    return np.float32((one - xx) * (one - yy) * d[yf, xf] + xx * (one - yy) * d[yf, xf + 1] +
                      (one - xx) * yy * d[yf + 1, xf] + xx * yy * d[yf + 1, xf + 1])


def test_zero_iterations_reports_the_initial_errors():
    """max_iterations = 0: the pose is returned unchanged (through the float -> quaternion -> float round trip)
    and nothing is evaluated: every edge keeps level 0 / error 0 -> all inliers, as g2o would report."""
    sc = dust_scene.make_scene(3)
    r = oracle.align_dust(sc["dust"], sc["pts"], sc["Tcw_init"], sc["fx"], sc["fy"], sc["cx"], sc["cy"], max_iterations=0)
    assert r["iterations"] == 0 and r["n_inlier"] == len(sc["pts"])
    assert np.abs(r["Tcw"] - sc["Tcw_init"]).max() < 1e-6


def test_converges_towards_the_true_pose_and_flags_outliers():
    for seed in range(6):
        sc = dust_scene.make_scene(seed)
        r = oracle.align_dust(sc["dust"], sc["pts"], sc["Tcw_init"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
        uv_t, _ = dust_scene.project(sc["Tcw_true"], sc["pts"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
        uv_0, _ = dust_scene.project(sc["Tcw_init"], sc["pts"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
        uv_1, _ = dust_scene.project(r["Tcw"], sc["pts"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
        n_out = 16
        e0, e1 = np.abs(uv_0 - uv_t)[n_out:].mean(), np.abs(uv_1 - uv_t)[n_out:].mean()
        assert e1 < 0.6 * e0, (seed, e0, e1)         # the alignment pulls the projections onto the keypoints
        assert 1 <= r["iterations"] <= 40
        assert r["n_inlier"] == int(r["inlier"].sum())
        # inlier rule (:258-270): chi2 = (bilinear dust at the final projection)^2 <= 0.9
        for i in np.flatnonzero(r["inlier"])[:40]:
            u, v = r["uv"][i]
            assert _bilinear(sc["dust"], u, v) ** 2 <= 0.9 + 1e-6
        # the flags against an independent evaluation of the rule at the returned pose (a few may differ: the
        # edges keep the errors of the LAST trial, which may be a rejected one, and `level` is sticky)
        hc, wc = sc["dust"].shape
        uv, z = dust_scene.project(r["Tcw"], sc["pts"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
        mism = 0
        for i in range(len(uv)):
            u, v = uv[i]
            inside = z[i] > 0 and u >= 1 and u + 2 < wc and v >= 1 and v + 2 < hc
            chi2 = float(_bilinear(sc["dust"], u, v)) ** 2 if inside else 9.0
            if abs(chi2 - 0.9) > 2e-2:
                mism += int(bool(r["inlier"][i]) != (chi2 <= 0.9))
        assert mism <= 3, (seed, mism)


def test_points_behind_the_camera_and_outside_the_map_are_never_inliers():
    sc = dust_scene.make_scene(7, n_points=64, outlier_frac=0.0)
    pts = sc["pts"].copy()
    T = sc["Tcw_init"].astype(np.float64)
    Rt, t = T[:3, :3], T[:3, 3]
    behind = (np.array([0.1, -0.2, -3.0]) - t) @ Rt          # z < 0 in the camera frame (:70-76)
    far_left = (np.array([-30.0, 0.0, 4.0]) - t) @ Rt        # projects left of the map (:84-87)
    pts[0], pts[1] = behind, far_left
    r = oracle.align_dust(sc["dust"], pts, sc["Tcw_init"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
    assert not r["inlier"][0] and not r["inlier"][1]
    assert r["inlier"][2:].sum() > 40


def test_a_perfect_start_on_a_flat_map_terminates_at_once():
    """A constant dust map has zero gradient: H = 0, lambda = tau * 0 = 0, the solver reports 'not positive',
    ten rejected trials -> Terminate after the first iteration (OptimizationAlgorithmLevenberg::solve)."""
    sc = dust_scene.make_scene(2, n_points=32)
    flat = np.full_like(sc["dust"], 0.5)
    r = oracle.align_dust(flat, sc["pts"], sc["Tcw_true"], sc["fx"], sc["fy"], sc["cx"], sc["cy"])
    assert r["iterations"] == 1
    assert np.abs(r["Tcw"] - sc["Tcw_true"]).max() < 1e-6
    assert r["n_inlier"] == int(r["inlier"].sum()) and r["inlier"].sum() >= 28   # 0.25 <= 0.9 where in view
