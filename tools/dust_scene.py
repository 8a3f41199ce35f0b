"""Synthetic scenes for the dust-alignment path (tests, bench): a camera, map points in front of it, and a
dust map (softmax dustbin probability per 8x8 cell: ~1 where there is no keypoint, low where there is one)
whose minima sit at the map points' true projections.  Seeded, numpy only."""
import numpy as np


def make_scene(seed, H=480, W=752, n_points=160, fx=458.654, fy=457.296, cx=367.215, cy=248.375, sigma_cells=1.6,
               rot_deg=1.0, trans=0.04, outlier_frac=0.1):
    """Returns dict(dust [H/8, W/8] f32, pts [n, 3] f32, Tcw_true, Tcw_init [4, 4] f32, fx, fy, cx, cy)."""
    rng = np.random.default_rng(seed)
    hc, wc = H // 8, W // 8
    # true pose: small rotation + translation from identity
    def rot(rx, ry, rz):
        cxr, sxr, cyr, syr, czr, szr = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
        Rx = np.array([[1, 0, 0], [0, cxr, -sxr], [0, sxr, cxr]])
        Ry = np.array([[cyr, 0, syr], [0, 1, 0], [-syr, 0, cyr]])
        Rz = np.array([[czr, -szr, 0], [szr, czr, 0], [0, 0, 1]])
        return Rz @ Ry @ Rx
    T_true = np.eye(4)
    T_true[:3, :3] = rot(*(rng.uniform(-0.2, 0.2, 3)))
    T_true[:3, 3] = rng.uniform(-0.5, 0.5, 3)
    # points: sample pixels (away from the border), depths, back-project into the world
    u = rng.uniform(40, W - 40, n_points)
    v = rng.uniform(40, H - 40, n_points)
    z = rng.uniform(2.0, 9.0, n_points)
    pc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
    Rt, tt = T_true[:3, :3], T_true[:3, 3]
    pw = (pc - tt) @ Rt          # R^T (pc - t)
    n_out = int(outlier_frac * n_points)
    # dust map: 1 - sum of Gaussians at the inlier points' projections in cell coordinates
    uc = (u - 3.5) / 8.0
    vc = (v - 3.5) / 8.0
    yy, xx = np.mgrid[0:hc, 0:wc]
    keyp = np.zeros((hc, wc))
    for i in range(n_out, n_points):
        keyp = np.maximum(keyp, np.exp(-((xx - uc[i]) ** 2 + (yy - vc[i]) ** 2) / (2 * sigma_cells ** 2)))
    dust = (0.97 - 0.9 * keyp + 0.01 * rng.standard_normal((hc, wc))).clip(0.01, 0.99).astype(np.float32)
    # outliers: world points whose projection has no keypoint nearby (moved far away in the image)
    pw[:n_out] += rng.uniform(1.0, 2.0, (n_out, 3)) * rng.choice([-1, 1], (n_out, 3))
    # initial pose: the true one perturbed
    dT = np.eye(4)
    dT[:3, :3] = rot(*(np.deg2rad(rot_deg) * rng.uniform(-1, 1, 3)))
    dT[:3, 3] = trans * rng.uniform(-1, 1, 3)
    T_init = dT @ T_true
    return dict(dust=dust, pts=pw.astype(np.float32), Tcw_true=T_true.astype(np.float32),
                Tcw_init=T_init.astype(np.float32), fx=np.float32(fx), fy=np.float32(fy), cx=np.float32(cx),
                cy=np.float32(cy))


def project(Tcw, pts, fx, fy, cx, cy):
    """Dust-map (cell) coordinates of world points under pose Tcw, as optimizer_dust.cpp:223-226 scales them."""
    T = np.asarray(Tcw, np.float64)
    p = np.asarray(pts, np.float64) @ T[:3, :3].T + T[:3, 3]
    return np.stack([p[:, 0] * (fx / 8.0) / p[:, 2] + (cx - 3.5) / 8.0, p[:, 1] * (fy / 8.0) / p[:, 2] + (cy - 3.5) / 8.0], 1), p[:, 2]
