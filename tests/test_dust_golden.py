"""The dust alignment against INDEPENDENT fixtures (tests/golden/dust_*.npz, made by tests/golden/make_golden_dust.py:
an f64 numpy / scipy statement of types_dust_tracking.cpp:37-140 and optimizer_dust.cpp:170-294 that shares no code
with include/spfe_dust_math.h — expm for the exponential map, 4x4 matrices for poses, scipy's solver).

CPU suite: the oracle (which compiles include/spfe_dust_math.h, the product's arithmetic) reproduces the fixtures:
pose <= 1e-9 in double precision, iteration count, inlier flags, dust_proj_u / v, the per-edge error / level / 1x6
Jacobian at the start pose; spfe_dust_jacobian against NUMERIC derivatives (<= 1e-6); spfe_se3_oplus against expm.
GPU suite: spfe_align_dust does the same (its pose output is the float 4x4 of Frame::mTcw: <= 1 float ulp of the
fixture's pose rounded to float)."""
import glob
import os

import numpy as np
import pytest
import scipy.linalg

from oracle import oracle

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "dust_*.npz")))
IDS = [os.path.basename(p)[5:-4] for p in GOLD]

POSE64_TOL = 1e-9       # double-precision pose, oracle vs fixture
UV_TOL = 2e-6           # dust_proj_u / v are floats of magnitude <= 160: one ulp is 1.5e-5 at 128; relative bound below
J_TOL = 1e-6            # analytic vs numeric / independent Jacobian (relative to max(1, |J|))


def _load(path):
    z = np.load(path)
    return {k: z[k] for k in z.files}


def test_fixture_set_covers_what_the_verdict_asked_for():
    assert len(GOLD) >= 6
    g = {i: _load(p) for i, p in zip(IDS, GOLD)}
    assert (g["behind_outside"]["level0"][:3] == 1).all()                                  # z < 0, outside at the start
    assert ((g["leaving"]["level0"] == 0) & (g["leaving"]["level"] == 1)).sum() >= 10      # points leave during the trials
    assert g["flat"]["iterations"] == 1 and g["std2"]["iterations"] == 40
    assert all(int(v["rejected"]) > 0 for v in g.values())                                 # every scene has rejected steps


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_oracle_reproduces_independent_fixture(path):
    g = _load(path)
    fx, fy, cx, cy = g["intr"]
    r = oracle.align_dust(g["dust"], g["pts"], g["Tcw_init"], fx, fy, cx, cy, max_iterations=int(g["max_iterations"]),
                          delta=float(g["delta"]))
    assert r["iterations"] == int(g["iterations"])
    assert np.abs(r["pose64"] - g["pose64"]).max() <= POSE64_TOL
    assert np.abs(r["Tcw"].astype(np.float64) - g["Tcw"]).max() <= 1.2e-7 * max(1.0, np.abs(g["Tcw"]).max())
    assert np.array_equal(r["inlier"], g["inlier"]) and r["n_inlier"] == int(g["n_inlier"])
    seen = g["level"] == 0                    # u_, v_ of an edge that was never inside the map are uninitialised
    assert np.abs(r["uv"][seen] - g["uv"][seen]).max(initial=0) <= UV_TOL * 160


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_edge_error_level_and_jacobian_at_the_start_pose(path):
    """spfe_dust_error / spfe_dust_jacobian (include/spfe_dust_math.h:180-226, through the oracle's test hook) against
    the fixture's independent evaluation of computeError / linearizeOplus for every map point of the scene."""
    g = _load(path)
    fx, fy, cx, cy = g["intr"]
    for i, X in enumerate(g["pts"]):
        e = oracle.dust_edge(g["dust"], X, g["Tcw_init"], fx, fy, cx, cy)
        assert e["level"] == int(g["level0"][i])
        assert abs(e["err"] - g["err0"][i]) <= 1e-12        # the same float bilinear value
        assert np.abs(e["J"] - g["J0"][i]).max() <= 1e-9 * max(1.0, np.abs(g["J0"][i]).max())
        if i == 0:
            assert np.abs(e["pose64"] - g["pose64_init"]).max() <= 1e-12      # Converter::toSE3Quat


def _twist(u):
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]]
    M[:3, 3] = u[3:]
    return M


def test_oplus_is_the_matrix_exponential():
    """spfe_se3_oplus (SE3Quat::exp(update) * T, include/spfe_dust_math.h:87) against scipy.linalg.expm of the 4x4 twist,
    over small (Taylor branch), ordinary and large updates."""
    g = _load(GOLD[IDS.index("std0")])
    fx, fy, cx, cy = g["intr"]
    rng = np.random.default_rng(1)
    for s in (1e-8, 1e-6, 1e-4, 1e-2, 0.3, 2.5):
        u = s * rng.standard_normal(6)
        e = oracle.dust_edge(g["dust"], g["pts"][0], g["Tcw_init"], fx, fy, cx, cy, update=u)
        want = scipy.linalg.expm(_twist(u)) @ g["pose64_init"]
        assert np.abs(e["pose64"] - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), s


def _plane_map(hc, wc, a, b, c):
    yy, xx = np.mgrid[0:hc, 0:wc]
    return (a * xx + b * yy + c).astype(np.float32)


def test_analytic_jacobian_against_numeric_derivatives():
    """linearizeOplus (types_dust_tracking.cpp:100-140) = [pixel gradient by +-1 cell differences] x [d(u, v) / d(twist)].
    (1) the 2x6 projection part: on the maps dust = x and dust = y the bilinear lookup IS the projection, so
        spfe_dust_jacobian returns d u / d xi and d v / d xi; compared with central differences of an independent f64
        projection under expm perturbations: <= 1e-6.
    (2) on a general plane a x + b y + c the +-1 differences are the true gradient: J = numeric derivative of the
        oracle's own error under spfe_se3_oplus perturbations (float lookups: 2e-4 of the gradient's scale)."""
    g = _load(GOLD[IDS.index("std1")])
    fx, fy, cx, cy = g["intr"]
    hc, wc = g["dust"].shape
    K = (float(np.float32(fx) / np.float32(8)), float(np.float32(fy) / np.float32(8)), (float(cx) - 3.5) / 8.0, (float(cy) - 3.5) / 8.0)
    T0 = g["pose64_init"]

    def proj(T, X):
        p = T[:3, :3] @ X + T[:3, 3]
        return np.array([p[0] * K[0] / p[2] + K[2], p[1] * K[1] / p[2] + K[3]])

    mx, my = _plane_map(hc, wc, 1, 0, 0), _plane_map(hc, wc, 0, 1, 0)
    plane = _plane_map(hc, wc, 0.004, -0.007, 0.4)
    checked = 0
    for X in g["pts"][16:80]:
        X64 = X.astype(np.float64)
        u, v = proj(T0, X64)
        if not (2 <= u < wc - 3 and 2 <= v < hc - 3):
            continue
        Ju = oracle.dust_edge(mx, X, g["Tcw_init"], fx, fy, cx, cy)["J"]
        Jv = oracle.dust_edge(my, X, g["Tcw_init"], fx, fy, cx, cy)["J"]
        Jn = np.zeros((2, 6))
        h = 1e-6
        for k in range(6):
            d = np.zeros(6)
            d[k] = h
            Jn[:, k] = (proj(scipy.linalg.expm(_twist(d)) @ T0, X64) - proj(scipy.linalg.expm(_twist(-d)) @ T0, X64)) / (2 * h)
        scale = max(1.0, np.abs(Jn).max())
        # the +-1 differences of the maps x / y are (1, 0) / (0, 1) up to the float rounding of the lookups (values up to
        # 94: 1e-5), so the raw rows agree to 2e-5 ...
        assert np.abs(np.stack([Ju, Jv]) - Jn).max() <= 2e-5 * scale
        # ... and with the 2x2 gradient factor G taken out exactly — [J_x; J_y] = G * Juv, G read off the translation
        # columns, where Juv[:, 3:5] = diag(fx / z, fy / z) — the other eight entries of the analytic 2x6 agree with the
        # numeric derivative to 1e-6
        z = (T0[:3, :3] @ X64 + T0[:3, 3])[2]
        G = np.array([[Ju[3] / (K[0] / z), Ju[4] / (K[1] / z)], [Jv[3] / (K[0] / z), Jv[4] / (K[1] / z)]])
        assert np.abs(G - np.eye(2)).max() <= 1e-4
        Juv = np.linalg.solve(G, np.stack([Ju, Jv]))
        assert np.abs(Juv - Jn).max() <= J_TOL * scale
        # (2) total derivative on a plane
        e0 = oracle.dust_edge(plane, X, g["Tcw_init"], fx, fy, cx, cy)
        hh = 1e-3
        Jt = np.zeros(6)
        for k in range(6):
            d = np.zeros(6)
            d[k] = hh
            Jt[k] = (oracle.dust_edge(plane, X, g["Tcw_init"], fx, fy, cx, cy, update=d)["err"] -
                     oracle.dust_edge(plane, X, g["Tcw_init"], fx, fy, cx, cy, update=-d)["err"]) / (2 * hh)
        assert np.abs(e0["J"] - Jt).max() <= 2e-4 * max(1.0, np.abs(Jt).max())
        checked += 1
    assert checked >= 40


def test_huber_and_solver_pieces():
    """RobustKernelHuber and the LM bookkeeping are exercised end to end by the fixtures (tight delta 0.3, rejected steps
    in every scene); here: the zero system (flat map) stops after one iteration with the pose untouched."""
    g = _load(GOLD[IDS.index("flat")])
    fx, fy, cx, cy = g["intr"]
    r = oracle.align_dust(g["dust"], g["pts"], g["Tcw_init"], fx, fy, cx, cy)
    assert r["iterations"] == 1 and np.abs(r["pose64"] - g["pose64_init"]).max() <= 1e-15


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_kernel_reproduces_independent_fixture(path):
    from sp_orb_slam_amd import weights
    from sp_orb_slam_amd.extractor import SPExtractor
    g = _load(path)
    fx, fy, cx, cy = g["intr"]
    hc, wc = g["dust"].shape
    ext = SPExtractor(100, hc * 8, wc * 8, weights.synthetic(7, "dense"), with_heat=False)
    r = ext.align_dust(g["dust"], g["pts"], g["Tcw_init"], fx, fy, cx, cy, max_iterations=int(g["max_iterations"]),
                       huber_delta=float(g["delta"]))
    ext.close()
    assert r["iterations"] == int(g["iterations"])
    assert np.abs(r["Tcw"].astype(np.float64) - g["pose64"]).max() <= 1.2e-7 * max(1.0, np.abs(g["pose64"]).max())
    assert np.array_equal(r["inlier"], g["inlier"]) and r["n_inlier"] == int(g["n_inlier"])
    seen = g["level"] == 0
    assert np.abs(r["uv"][seen] - g["uv"][seen]).max(initial=0) <= UV_TOL * 160


@pytest.mark.gpu
def test_kernel_record_and_batch_forms_reproduce_the_fixtures():
    """spfe_align_dust_batch_device: the fixtures of one map size as ONE launch (a workgroup per frame), each frame's
    dense_dust planted in a device record; every frame must land on its fixture."""
    import torch
    from sp_orb_slam_amd import parallel, weights
    from sp_orb_slam_amd.extractor import DUST_OUT_BYTES, SPExtractor
    sel = [_load(p) for p in GOLD if _load(p)["dust"].shape == (60, 94) and int(_load(p)["max_iterations"]) == 40]
    assert len(sel) >= 6
    ext = SPExtractor(100, 480, 752, weights.synthetic(7, "dense"), with_heat=False)
    lay = parallel.RecordLayout(480, 752, 100)
    rb = ext.record_bytes()
    nb = len(sel)
    recs = torch.zeros(nb * rb, dtype=torch.uint8, device="cuda")
    pts = torch.zeros((nb, 512, 3), dtype=torch.float32, device="cuda")
    npts = torch.zeros(nb, dtype=torch.int32, device="cuda")
    Tin = torch.zeros((nb, 16), dtype=torch.float32, device="cuda")
    for f, g in enumerate(sel):
        off = f * rb + lay.off_dd
        recs[off:off + 60 * 94 * 4] = torch.from_numpy(g["dust"].reshape(-1).view(np.uint8).copy()).cuda()
        n = len(g["pts"])
        pts[f, :n] = torch.from_numpy(g["pts"]).cuda()
        npts[f] = n
        Tin[f] = torch.from_numpy(g["Tcw_init"].reshape(16)).cuda()
    out = torch.zeros(nb * DUST_OUT_BYTES, dtype=torch.uint8, device="cuda")
    fx, fy, cx, cy = sel[0]["intr"]
    assert all(np.array_equal(g["intr"], sel[0]["intr"]) for g in sel)
    ext.align_dust_batch_device(recs.data_ptr(), nb, pts.data_ptr(), npts.data_ptr(), Tin.data_ptr(), out.data_ptr(), fx, fy, cx, cy)
    torch.cuda.synchronize()
    raw = out.cpu().numpy()
    for f, g in enumerate(sel):
        r = ext.decode_dust_out(raw[f * DUST_OUT_BYTES:(f + 1) * DUST_OUT_BYTES], len(g["pts"]))
        assert r["iterations"] == int(g["iterations"])
        assert np.abs(r["Tcw"].astype(np.float64) - g["pose64"]).max() <= 1.2e-7 * max(1.0, np.abs(g["pose64"]).max())
        assert np.array_equal(r["inlier"], g["inlier"])
    ext.close()
