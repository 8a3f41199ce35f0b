"""Fixtures for the direct "dust" alignment: an INDEPENDENT f64 numpy / scipy statement of

    g2o::EdgeSE3ProjectDustOnlyPose   /root/reference/orb_slam2/src/optimization/types_dust_tracking.cpp:37-140
    Optimizer::PoseOptimizationDust   /root/reference/orb_slam2/src/mapping/optimizer_dust.cpp:170-294
    Converter::toSE3Quat / toCvMat    /root/reference/orb_slam2/src/utils/converter.cpp:36-67

and of the g2o pieces they drive (g2o itself is a catkin dependency, not in /root/reference; its published
algorithm is what is stated: SparseOptimizer::optimize, OptimizationAlgorithmLevenberg::solve with tau = 1e-5,
the [1/3, 2/3] good-step scale, ni doubling, 10 trials after a failure; BaseUnaryEdge::constructQuadraticForm;
RobustKernelHuber; VertexSE3Expmap::oplusImpl = SE3Quat::exp(update) * estimate; LinearSolverDense).

It shares NO code with include/spfe_dust_math.h, oracle/ or the kernels: poses are 4x4 double matrices, the
exponential map is scipy.linalg.expm of the 4x4 twist, the linear solve is scipy's, sums are Python loops in
edge order.  Run in the build container only:

    python tests/golden/make_golden_dust.py        -> tests/golden/dust_*.npz

The script checks itself while it runs: the reference's analytic 2x6 projection Jacobian against central
differences of the projection under expm perturbations (<= 1e-6), and SE3Quat::exp's closed form against expm.
tests/test_dust_golden.py holds the oracle (CPU suite) and the kernel (GPU suite) to these files.
"""
import os
import sys

import numpy as np
import scipy.linalg

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from tools import dust_scene  # noqa: E402  (scene generator only: points, poses, a dust map)

F = np.float32


# ---- pose conversions (converter.cpp:36-67) -----------------------------------------------------------------------

def quat_from_matrix_eigen(R):
    """Eigen::Quaterniond(Matrix3d) (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<.., 3, 3>):
    the trace branch, else the largest-diagonal branch.  Returns (x, y, z, w)."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[2, 1] - R[1, 2]) * t
        q[1] = (R[0, 2] - R[2, 0]) * t
        q[2] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (R[k, j] - R[j, k]) * t
        q[j] = (R[j, i] + R[i, j]) * t
        q[k] = (R[k, i] + R[i, k]) * t
    return q


def matrix_from_quat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def pose_from_cvmat(Tcw32):
    """toSE3Quat: float R, t -> double; SE3Quat(R, t) keeps a NORMALISED quaternion (w >= 0), so a float matrix that
    is orthonormal only to 1e-7 is projected onto a rotation here — the matrix is not used as it is."""
    T32 = np.asarray(Tcw32, np.float32).reshape(4, 4).astype(np.float64)
    q = quat_from_matrix_eigen(T32[:3, :3])
    if q[3] < 0:
        q = -q
    q = q / np.sqrt(q @ q)
    T = np.eye(4)
    T[:3, :3] = matrix_from_quat(q)
    T[:3, 3] = T32[:3, 3]
    return T


def twist_matrix(upd):
    """update = (omega, upsilon) -> the 4x4 element of se(3) whose expm is SE3Quat::exp(update)."""
    wx, wy, wz = upd[:3]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -wz, wy], [wz, 0, -wx], [-wy, wx, 0]]
    M[:3, 3] = upd[3:]
    return M


def oplus(T, upd):
    """VertexSE3Expmap::oplusImpl: estimate <- SE3Quat::exp(update) * estimate."""
    return scipy.linalg.expm(twist_matrix(upd)) @ T


def se3quat_exp_closed_form(upd):
    """SE3Quat::exp as g2o writes it (Rodrigues + V), for the self-check against expm only."""
    om, ups = np.asarray(upd[:3], float), np.asarray(upd[3:], float)
    th = np.sqrt(om @ om)
    Om = twist_matrix(upd)[:3, :3]
    Om2 = Om @ Om
    if th < 0.00001:
        R = np.eye(3) + Om + 0.5 * Om2
        V = np.eye(3) + 0.5 * Om + Om2 / 6.0
    else:
        R = np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / th ** 2 * Om2
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * Om2
    E = np.eye(4)
    E[:3, :3] = R
    E[:3, 3] = V @ ups
    return E


# ---- the edge (types_dust_tracking.cpp:37-140) ----------------------------------------------------------------------

def pixel_value(dust, x, y):
    """getPixelValue(float x, float y) :44-58 — float arithmetic, left to right."""
    x, y = F(x), F(y)
    xf, yf = int(np.floor(x)), int(np.floor(y))
    xx, yy = F(x - F(xf)), F(y - F(yf))
    one = F(1)
    return F(F(F(F(one - xx) * F(one - yy)) * dust[yf, xf]) + F(F(xx * F(one - yy)) * dust[yf, xf + 1]) +
             F(F(F(one - xx) * yy) * dust[yf + 1, xf]) + F(F(xx * yy) * dust[yf + 1, xf + 1]))


def in_image(u, v, w, h, border=1.0):
    """isInImage :37-42; w_, h_ are floats."""
    return bool(u >= border and u + border + 1 < F(w) and v >= border and v + border + 1 < F(h))


class Edge:
    def __init__(self, Xw):
        self.Xw = np.asarray(Xw, np.float32).astype(np.float64)    # e->Xw[k] = Xw.at<float>(k)
        self.err = 0.0
        self.u = F(0)
        self.v = F(0)
        self.level = 0                                             # setLevel(1) is never undone

    def compute_error(self, T, K, dust):
        fx, fy, cx, cy = K
        p = T[:3, :3] @ self.Xw + T[:3, 3]
        if p[2] < 0.0:
            self.err, self.level = 0.0, 1
            return
        x = p[0] * fx / p[2] + cx
        y = p[1] * fy / p[2] + cy
        hc, wc = dust.shape
        if not in_image(x, y, wc, hc):
            self.err, self.level = 0.0, 1
        else:
            self.err = float(pixel_value(dust, x, y))
            self.u, self.v = F(x), F(y)

    def projection_jacobian(self, T, K):
        """jacobian_uv_ksai :118-131 (2x6, columns: rotation then translation)."""
        fx, fy, cx, cy = K
        x, y, z = T[:3, :3] @ self.Xw + T[:3, 3]
        iz = 1.0 / z
        iz2 = iz * iz
        Juv = np.array([[-x * y * iz2 * fx, (1 + x * x * iz2) * fx, -y * iz * fx, iz * fx, 0.0, -x * iz2 * fx],
                        [-(1 + y * y * iz2) * fy, x * y * iz2 * fy, x * iz * fy, 0.0, iz * fy, -y * iz2 * fy]])
        return Juv, x * fx * iz + cx, y * fy * iz + cy

    def linearize(self, T, K, dust):
        if self.level == 1:
            return np.zeros(6)
        Juv, u, v = self.projection_jacobian(T, K)
        hc, wc = dust.shape
        if not in_image(u, v, wc, hc):
            raise RuntimeError(" should be omitted")
        gu = float(F(pixel_value(dust, u + 1, v) - pixel_value(dust, u - 1, v)) / F(2.0))
        gv = float(F(pixel_value(dust, u, v + 1) - pixel_value(dust, u, v - 1)) / F(2.0))
        return np.array([gu, gv]) @ Juv


def project(T, Xw, K):
    fx, fy, cx, cy = K
    p = T[:3, :3] @ Xw + T[:3, 3]
    return np.array([p[0] * fx / p[2] + cx, p[1] * fy / p[2] + cy])


def huber(e2, delta):
    """RobustKernelHuber::robustify."""
    d2 = delta * delta
    if e2 <= d2:
        return e2, 1.0
    s = np.sqrt(e2)
    return 2 * s * delta - d2, delta / s


def intrinsics(fx, fy, cx, cy):
    """optimizer_dust.cpp:223-226: fx / 8.0f is float / float; (cx - 3.5) / 8.0f is double."""
    return (float(F(fx) / F(8.0)), float(F(fy) / F(8.0)), (float(F(cx)) - 3.5) / 8.0, (float(F(cy)) - 3.5) / 8.0)


# ---- the Levenberg loop ---------------------------------------------------------------------------------------------

def active_chi2(edges, T, K, dust, delta):
    """computeActiveErrors + activeRobustChi2 (edges in insertion order)."""
    chi = 0.0
    for e in edges:
        e.compute_error(T, K, dust)
        chi += huber(e.err * e.err, delta)[0]
    return chi


def solve_dense(H, lam, b):
    """LinearSolverDense: Eigen::LDLT of H + lambda I; fails when not positive semi-definite.  A zero matrix counts as
    positive there and its solution is 0."""
    A = H + lam * np.eye(6)
    if not np.any(A):
        return True, np.zeros(6)
    try:
        c = scipy.linalg.cho_factor(A)
    except np.linalg.LinAlgError:
        return False, np.zeros(6)
    return True, scipy.linalg.cho_solve(c, b)


def pose_optimization_dust(dust, pts, Tcw32, fx, fy, cx, cy, max_iterations=40, delta=0.9, inlier_chi2=0.9,
                           trace=None):
    dust = np.asarray(dust, np.float32)
    K = intrinsics(fx, fy, cx, cy)
    T = pose_from_cvmat(Tcw32)
    edges = [Edge(p) for p in np.asarray(pts, np.float32).reshape(-1, 3)]
    lam, ni = 0.0, 2.0
    iterations, ok = 0, True
    it = 0
    while it < max_iterations and ok:
        current = active_chi2(edges, T, K, dust, delta)
        H, b = np.zeros((6, 6)), np.zeros(6)
        for e in edges:                                       # buildSystem
            J = e.linearize(T, K, dust)
            w = huber(e.err * e.err, delta)[1]
            b -= w * J * e.err
            H += np.outer(J * w, J)
        if it == 0:
            lam, ni = 1e-5 * np.abs(np.diag(H)).max(initial=0.0), 2.0
        rho, qmax = 0.0, 0
        while True:
            saved = T.copy()
            ok2, x = solve_dense(H, lam, b)
            T = oplus(T, x)
            temp = active_chi2(edges, T, K, dust, delta)
            if not ok2:
                temp = np.finfo(np.float64).max
            scale = float(x @ (lam * x + b)) + 1e-3
            rho = (current - temp) / scale
            accepted = rho > 0 and np.isfinite(temp)
            if trace is not None:
                trace.append((it, qmax, current, temp, rho, lam, accepted))
            if accepted:
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                ni = 2.0
                current = temp
            else:
                lam *= ni
                ni *= 2
                T = saved
            qmax += 1
            if not (rho < 0 and qmax < 10):
                break
        iterations += 1
        it += 1
        if qmax == 10 or rho == 0:
            ok = False
    inlier = np.array([not (e.level == 1 or e.err * e.err > inlier_chi2) for e in edges], bool)
    uv = np.array([[e.u, e.v] for e in edges], np.float32).reshape(-1, 2)
    # keep the rotation a rotation the way SE3Quat does (quaternion normalised after every product); expm products drift
    # by ~1e-16 per step, so this changes nothing above 1e-15 — stated for completeness
    U, _, Vt = np.linalg.svd(T[:3, :3])
    T[:3, :3] = U @ Vt
    return dict(pose64=T, Tcw=T.astype(np.float32), inlier=inlier, uv=uv, n_inlier=int(inlier.sum()),
                iterations=iterations, level=np.array([e.level for e in edges], np.int32),
                err=np.array([e.err for e in edges]))


# ---- self checks ----------------------------------------------------------------------------------------------------

def numeric_projection_jacobian(T, Xw, K, h=1e-6):
    J = np.zeros((2, 6))
    for k in range(6):
        d = np.zeros(6)
        d[k] = h
        J[:, k] = (project(oplus(T, d), Xw, K) - project(oplus(T, -d), Xw, K)) / (2 * h)
    return J


def self_check(sc):
    K = intrinsics(sc["fx"], sc["fy"], sc["cx"], sc["cy"])
    T = pose_from_cvmat(sc["Tcw_init"])
    worst = 0.0
    for p in sc["pts"][:24]:
        e = Edge(p)
        Ja, _, _ = e.projection_jacobian(T, K)
        Jn = numeric_projection_jacobian(T, e.Xw, K)
        worst = max(worst, np.abs(Ja - Jn).max() / max(1.0, np.abs(Jn).max()))
    assert worst <= 1e-6, worst
    rng = np.random.default_rng(0)
    for s in (1e-7, 1e-3, 0.3, 2.0):
        u = s * rng.standard_normal(6)
        assert np.abs(se3quat_exp_closed_form(u) - scipy.linalg.expm(twist_matrix(u))).max() <= 1e-12
    return worst


# ---- scenes ----------------------------------------------------------------------------------------------------------

def scenes():
    out = []
    for seed in range(3):
        out.append(("std%d" % seed, dust_scene.make_scene(seed), {}))
    out.append(("vga200", dust_scene.make_scene(3, H=480, W=640, n_points=200, cx=311.2, cy=248.4), {}))
    # z < 0 and outside the map from the start (types_dust_tracking.cpp:70-76, :84-87)
    sc = dust_scene.make_scene(7, n_points=64, outlier_frac=0.0)
    T = sc["Tcw_init"].astype(np.float64)
    sc["pts"][0] = ((np.array([0.1, -0.2, -3.0]) - T[:3, 3]) @ T[:3, :3]).astype(np.float32)
    sc["pts"][1] = ((np.array([-30.0, 0.0, 4.0]) - T[:3, 3]) @ T[:3, :3]).astype(np.float32)
    sc["pts"][2] = ((np.array([0.0, 9.0, 2.0]) - T[:3, 3]) @ T[:3, :3]).astype(np.float32)
    out.append(("behind_outside", sc, {}))
    # points leave the map DURING the trials (level 1 is sticky, :72-76 / :85-87): 40 map points that start 1.3 ... 3 cells
    # inside the map's edge under the INITIAL pose; 21 of them are pushed out by some trial step
    sc = dust_scene.make_scene(9, n_points=120, rot_deg=3.0, trans=0.2, outlier_frac=0.2)
    rng = np.random.default_rng(9)
    Ti = sc["Tcw_init"].astype(np.float64)
    for i in range(20, 60):
        side = rng.integers(0, 4)
        if side == 0:
            u, v = 8 * 1.3 + 3.5 + rng.uniform(0, 6), rng.uniform(30, 450)
        elif side == 1:
            u, v = 752 - 8 * 2.3 - 3.5 - rng.uniform(0, 6) + 3.5, rng.uniform(30, 450)
        elif side == 2:
            u, v = rng.uniform(30, 720), 8 * 1.3 + 3.5 + rng.uniform(0, 6)
        else:
            u, v = rng.uniform(30, 720), 480 - 8 * 2.3 - rng.uniform(0, 6)
        z = rng.uniform(2, 6)
        pc = np.array([(u - sc["cx"]) / sc["fx"] * z, (v - sc["cy"]) / sc["fy"] * z, z])
        sc["pts"][i] = ((pc - Ti[:3, 3]) @ Ti[:3, :3]).astype(np.float32)
    out.append(("leaving", sc, {}))
    # flat map: H = 0, lambda = 0 -> one iteration, Terminate (rho == 0)
    sc = dust_scene.make_scene(2, n_points=32)
    sc["dust"] = np.full_like(sc["dust"], 0.5)
    sc["Tcw_init"] = sc["Tcw_true"].copy()
    out.append(("flat", sc, {}))
    out.append(("small", dust_scene.make_scene(9, H=240, W=320, n_points=40, fx=230.0, fy=230.0, cx=158.0, cy=121.0), {}))
    out.append(("one_point", dust_scene.make_scene(10, n_points=1, outlier_frac=0.0), {}))
    out.append(("three_iters_tight_huber", dust_scene.make_scene(11, n_points=96), dict(max_iterations=3, delta=0.3)))
    return out


def main():
    for name, sc, kw in scenes():
        worst = self_check(sc) if len(sc["pts"]) >= 24 else 0.0
        trace = []
        r = pose_optimization_dust(sc["dust"], sc["pts"], sc["Tcw_init"], sc["fx"], sc["fy"], sc["cx"], sc["cy"],
                                   trace=trace, **kw)
        # the edge at the START pose: error, level and the full 1x6 Jacobian per point
        K = intrinsics(sc["fx"], sc["fy"], sc["cx"], sc["cy"])
        T0 = pose_from_cvmat(sc["Tcw_init"])
        e0, l0, J0 = [], [], []
        for p in sc["pts"]:
            e = Edge(p)
            e.compute_error(T0, K, sc["dust"])
            e0.append(e.err)
            l0.append(e.level)
            J0.append(e.linearize(T0, K, sc["dust"]))
        n_rej = sum(1 for t in trace if not t[-1])
        left = int(((np.array(l0) == 0) & (r["level"] == 1)).sum())
        path = os.path.join(HERE, "dust_%s.npz" % name)
        np.savez_compressed(
            path, dust=sc["dust"], pts=sc["pts"], Tcw_init=sc["Tcw_init"], intr=np.array([sc["fx"], sc["fy"], sc["cx"], sc["cy"]], np.float32),
            max_iterations=np.int32(kw.get("max_iterations", 40)), delta=np.float64(kw.get("delta", 0.9)),
            pose64=r["pose64"], Tcw=r["Tcw"], inlier=r["inlier"], uv=r["uv"], n_inlier=np.int32(r["n_inlier"]),
            iterations=np.int32(r["iterations"]), level=r["level"], err=r["err"], pose64_init=T0,
            err0=np.array(e0), level0=np.array(l0, np.int32), J0=np.array(J0).reshape(-1, 6),
            trials=np.int32(len(trace)), rejected=np.int32(n_rej))
        print("%-24s n=%3d it=%2d trials=%3d rejected=%2d inliers=%3d left_map=%2d  Jproj num-vs-analytic %.2e  %d B" %
              (name, len(sc["pts"]), r["iterations"], len(trace), n_rej, r["n_inlier"], left, worst, os.path.getsize(path)))


if __name__ == "__main__":
    main()
