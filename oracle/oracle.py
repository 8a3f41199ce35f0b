"""ctypes loader for the CPU oracle (oracle/spfe_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (sp_orb_slam_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libspfe_oracle.so")
_lib = None

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile oracle/libspfe_oracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "spfe_oracle.c")
    hdrs = [os.path.join(_HERE, "..", "include", h) for h in ("spfe_exact_math.h", "spfe_dust_math.h")]
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max([os.path.getmtime(src)] + [os.path.getmtime(h) for h in hdrs])):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libspfe_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.oracle_num_params.restype = C.c_size_t
        L.oracle_weight_offset.restype = C.c_size_t
        L.oracle_weight_offset.argtypes = [C.c_int]
        L.oracle_bias_offset.restype = C.c_size_t
        L.oracle_bias_offset.argtypes = [C.c_int]
        L.oracle_network.restype = C.c_int
        L.oracle_network.argtypes = [f32p, u8p, C.c_int, C.c_int, f32p, f32p, C.c_void_p]
        L.oracle_network_bf16.restype = C.c_int
        L.oracle_network_bf16.argtypes = [f32p, u8p, C.c_int, C.c_int, f32p, f32p]
        L.oracle_tail.restype = C.c_int
        L.oracle_tail.argtypes = [f32p, C.c_int, C.c_int, f32p, f32p, f32p, f32p, f32p, f32p, i32p]
        L.oracle_sample_desc.restype = None
        L.oracle_sample_desc.argtypes = [f32p, C.c_int, C.c_int, f32p, f32p, C.c_int, f32p]
        L.oracle_heat.restype = None
        L.oracle_heat.argtypes = [f32p, C.c_int, C.c_int, f32p, f32p, C.c_void_p]
        L.oracle_sort.restype = None
        L.oracle_sort.argtypes = [f32p, C.c_int, i32p]
        L.oracle_nms.restype = C.c_int
        L.oracle_nms.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 f32p, f32p, i32p, i16p]
        L.oracle_covariance.restype = None
        L.oracle_covariance.argtypes = [f32p, C.c_int, C.c_int, f32p, f32p, C.c_int, f32p, f32p,
                                        f32p]
        L.oracle_postprocess.restype = C.c_int
        L.oracle_postprocess.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p,
                                         f32p, f32p, f32p, i16p, f32p, f32p, f32p, f32p,
                                         C.POINTER(C.c_int)]
        L.oracle_match_bruteforce.restype = None
        L.oracle_match_bruteforce.argtypes = [f32p, C.c_int, f32p, C.c_int, C.c_int, i32p, f32p]
        L.oracle_match_patches.restype = None
        L.oracle_match_patches.argtypes = [f32p, f32p, C.c_int, i16p, C.c_int, C.c_int, f32p, C.c_int, C.c_float, i32p]
        L.oracle_stage_input.restype = C.c_int
        L.oracle_stage_input.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, u8p]
        L.oracle_set_num_threads.restype = None
        L.oracle_set_num_threads.argtypes = [C.c_int]
        L.oracle_get_max_threads.restype = C.c_int
        L.oracle_expf.restype = C.c_float
        L.oracle_expf.argtypes = [C.c_float]
        L.oracle_logf.restype = C.c_float
        L.oracle_logf.argtypes = [C.c_float]
        L.oracle_sum256.restype = C.c_float
        L.oracle_sum256.argtypes = [f32p]
        _lib = L
    return _lib


def num_params():
    return int(lib().oracle_num_params())


def network(blob, img):
    """u8 image [H,W] -> (semi [hc,wc,65], coarse [hc,wc,256], feat [hc,wc,128])."""
    H, W = img.shape
    hc, wc = H // 8, W // 8
    semi = np.empty((hc, wc, 65), np.float32)
    coarse = np.empty((hc, wc, 256), np.float32)
    feat = np.empty((hc, wc, 128), np.float32)
    rc = lib().oracle_network(np.ascontiguousarray(blob, np.float32),
                              np.ascontiguousarray(img, np.uint8), H, W, semi, coarse,
                              feat.ctypes.data)
    if rc:
        raise ValueError("oracle_network rc=%d" % rc)
    return semi, coarse, feat


def network_bf16(blob, img):
    """The build's bf16 mode (see oracle_network_bf16): -> (semi, coarse)."""
    H, W = img.shape
    hc, wc = H // 8, W // 8
    semi = np.empty((hc, wc, 65), np.float32)
    coarse = np.empty((hc, wc, 256), np.float32)
    rc = lib().oracle_network_bf16(np.ascontiguousarray(blob, np.float32),
                                   np.ascontiguousarray(img, np.uint8), H, W, semi, coarse)
    if rc:
        raise ValueError("oracle_network_bf16 rc=%d" % rc)
    return semi, coarse


def extract_bf16(blob, img, num_features):
    H, W = img.shape
    semi, coarse = network_bf16(blob, img)
    out = postprocess(semi, coarse, H, W, num_features)
    out["semi"] = semi
    out["coarse"] = coarse
    return out


def tail(semi, H, W):
    hc, wc = H // 8, W // 8
    dd = np.empty((hc, wc), np.float32)
    sd = np.empty((hc, wc), np.float32)
    hl = np.empty((H, W), np.float32)
    cx = np.empty(hc * wc, np.float32)
    cy = np.empty(hc * wc, np.float32)
    cs = np.empty(hc * wc, np.float32)
    cc = np.empty(hc * wc, np.int32)
    n = lib().oracle_tail(np.ascontiguousarray(semi, np.float32), H, W, dd, sd, hl, cx, cy, cs, cc)
    return dict(dense_dust=dd, semi_dust=sd, heat_log=hl, x=cx[:n].copy(), y=cy[:n].copy(),
                score=cs[:n].copy(), cell=cc[:n].copy())


def sample_desc(coarse, H, W, xs, ys):
    n = len(xs)
    out = np.empty((n, 256), np.float32)
    if n:
        lib().oracle_sample_desc(np.ascontiguousarray(coarse, np.float32), H, W,
                                 np.ascontiguousarray(xs, np.float32),
                                 np.ascontiguousarray(ys, np.float32), n, out)
    return out


def heat(heat_log):
    H, W = heat_log.shape
    h = np.empty((H, W), np.float32)
    hi = np.empty((H, W), np.float32)
    mm = np.zeros(2, np.float64)
    lib().oracle_heat(np.ascontiguousarray(heat_log, np.float32), H, W, h, hi, mm.ctypes.data)
    return h, hi, mm


def sort(score):
    n = len(score)
    order = np.empty(n, np.int32)
    lib().oracle_sort(np.ascontiguousarray(score, np.float32), n, order)
    return order


def nms(px, py, num_features, W, H, border=8, dist=4):
    n = len(px)
    cap = max(num_features + 2, 1)
    kx = np.empty(cap, np.float32)
    ky = np.empty(cap, np.float32)
    src = np.empty(cap, np.int32)
    occ = np.empty((H // 8, W // 8), np.int16)
    k = lib().oracle_nms(np.ascontiguousarray(px, np.float32), np.ascontiguousarray(py, np.float32),
                         n, num_features, border, dist, W, H, kx, ky, src, occ)
    return kx[:k].copy(), ky[:k].copy(), src[:k].copy(), occ


def covariance(heat_inv, kx, ky):
    H, W = heat_inv.shape
    k = len(kx)
    cov = np.empty((max(k, 1), 2), np.float32)
    cinv = np.empty((max(k, 1), 2), np.float32)
    resp = np.empty(max(k, 1), np.float32)
    lib().oracle_covariance(np.ascontiguousarray(heat_inv, np.float32), H, W,
                            np.ascontiguousarray(kx, np.float32),
                            np.ascontiguousarray(ky, np.float32), k, cov, cinv, resp)
    return cov[:k], cinv[:k], resp[:k]


def postprocess(semi, coarse, H, W, num_features):
    """SPExtractor::operator() after the network.  Returns a dict of all outputs."""
    hc, wc = H // 8, W // 8
    cap = num_features + 2
    kx = np.empty(cap, np.float32)
    ky = np.empty(cap, np.float32)
    resp = np.empty(cap, np.float32)
    desc = np.empty((cap, 256), np.float32)
    cov = np.empty((cap, 2), np.float32)
    cinv = np.empty((cap, 2), np.float32)
    occ = np.empty((hc, wc), np.int16)
    dd = np.empty((hc, wc), np.float32)
    sd = np.empty((hc, wc), np.float32)
    h = np.empty((H, W), np.float32)
    hi = np.empty((H, W), np.float32)
    ncand = C.c_int(0)
    k = lib().oracle_postprocess(np.ascontiguousarray(semi, np.float32),
                                 np.ascontiguousarray(coarse, np.float32), H, W, num_features, kx,
                                 ky, resp, desc, cov, cinv, occ, dd, sd, h, hi, C.byref(ncand))
    if k < 0:
        raise ValueError("oracle_postprocess rc=%d" % k)
    return dict(K=k, kp_xy=np.stack([kx[:k], ky[:k]], 1), response=resp[:k].copy(),
                desc=desc[:k].copy(), cov2=cov[:k].copy(), cov2_inv=cinv[:k].copy(), occ_grid=occ,
                dense_dust=dd, semi_dust=sd, heat=h, heat_inv=hi, n_candidates=ncand.value)


def extract(blob, img, num_features):
    """Full SPExtractor::operator(): u8 image -> dict of outputs (+ semi/coarse)."""
    if img is None or img.size == 0:
        raise RuntimeError("input image is empty")  # sp_extractor.cpp:364-365
    H, W = img.shape
    semi, coarse, _ = network(blob, img)
    out = postprocess(semi, coarse, H, W, num_features)
    out["semi"] = semi
    out["coarse"] = coarse
    return out


def match_bruteforce(query, train, cross_check=True):
    """cv::BFMatcher(NORM_L2, crossCheck).match as SPMatcher::SearchByBruteForce uses it
    (sp_matcher.cpp:1642-1674): -> (train_idx int32 [nq] with -1 = unmatched, distance f32 [nq])."""
    q = np.ascontiguousarray(query, np.float32).reshape(-1, 256)
    t = np.ascontiguousarray(train, np.float32).reshape(-1, 256)
    idx = np.empty(max(len(q), 1), np.int32)
    dist = np.empty(max(len(q), 1), np.float32)
    lib().oracle_match_bruteforce(q if len(q) else np.zeros((1, 256), np.float32), len(q),
                                  t if len(t) else np.zeros((1, 256), np.float32), len(t),
                                  1 if cross_check else 0, idx, dist)
    return idx[:len(q)].copy(), dist[:len(q)].copy()


def stage_input(src, H, W, map_x=None, map_y=None, rgb=False):
    """cv::remap(INTER_LINEAR) -> crop to H x W -> cvtColor(*2GRAY) on one raw frame
    (data_loader.cc:519-521, system.cpp:160-161, mono_tracker.cpp:18-28).  src: u8 [Hs,Ws] or [Hs,Ws,C]."""
    src = np.ascontiguousarray(src, np.uint8)
    cn = 1 if src.ndim == 2 else src.shape[2]
    hs, ws = src.shape[:2]
    gray = np.empty((H, W), np.uint8)
    mx = my = None
    if map_x is not None:
        mx = np.ascontiguousarray(map_x, np.float32)
        my = np.ascontiguousarray(map_y, np.float32)
        assert mx.shape == (hs, ws) and my.shape == (hs, ws)
    rc = lib().oracle_stage_input(src.reshape(-1), hs, ws, ws * cn, cn, 1 if rgb else 0,
                                  mx.ctypes.data if mx is not None else None,
                                  my.ctypes.data if my is not None else None, H, W, gray)
    if rc:
        raise ValueError("oracle_stage_input rc=%d" % rc)
    return gray


def set_num_threads(n):
    """OpenMP team size of the oracle's loops (timing legs only; results do not depend on it)."""
    lib().oracle_set_num_threads(int(n))
    return int(lib().oracle_get_max_threads())


def match_patches(mp_desc, mp_uv, occ_grid, kp_desc, max_dist=0.75):
    """Patch-wise association of projected map points (tracker_dust.cpp:113-172)."""
    m = np.ascontiguousarray(mp_desc, np.float32).reshape(-1, 256)
    uv = np.ascontiguousarray(mp_uv, np.float32).reshape(-1, 2)
    occ = np.ascontiguousarray(occ_grid, np.int16)
    kd = np.ascontiguousarray(kp_desc, np.float32).reshape(-1, 256)
    out = np.full(max(len(m), 1), -1, np.int32)
    if len(m):
        lib().oracle_match_patches(m, uv, len(m), occ, occ.shape[0], occ.shape[1],
                                   kd if len(kd) else np.zeros((1, 256), np.float32), len(kd), max_dist, out)
    return out[:len(m)].copy()


def align_dust(dust, pts, Tcw, fx, fy, cx, cy, max_iterations=40, delta=0.9, inlier_chi2=0.9):
    """Optimizer::PoseOptimizationDust (optimizer_dust.cpp:170-294) -> dict(Tcw, inlier, uv, n_inlier, iterations,
    pose64 = the pose in double precision before Converter::toCvMat's cast to float)."""
    dust = np.ascontiguousarray(dust, np.float32)
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
    Tin = np.ascontiguousarray(Tcw, np.float32).reshape(16)
    n = len(pts)
    Tout = np.zeros(16, np.float32)
    P64 = np.zeros(16, np.float64)
    inl = np.zeros(max(n, 1), np.uint8)
    uv = np.zeros((max(n, 1), 2), np.float32)
    it = C.c_int(0)
    L = lib()
    L.oracle_align_dust_pose64.restype = C.c_int
    L.oracle_align_dust_pose64.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_float,
                                           C.c_float, C.c_float, C.c_float, C.c_int, C.c_double, C.c_double, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
    k = L.oracle_align_dust_pose64(dust.ctypes.data, dust.shape[0], dust.shape[1], pts.ctypes.data, n, Tin.ctypes.data,
                                   float(fx), float(fy), float(cx), float(cy), int(max_iterations), float(delta),
                                   float(inlier_chi2), Tout.ctypes.data, inl.ctypes.data, uv.ctypes.data, C.byref(it),
                                   P64.ctypes.data)
    return dict(Tcw=Tout.reshape(4, 4), inlier=inl[:n].astype(bool), uv=uv[:n], n_inlier=int(k), iterations=it.value,
                pose64=P64.reshape(4, 4))


def dust_edge(dust, Xw, Tcw, fx, fy, cx, cy, update=None):
    """One EdgeSE3ProjectDustOnlyPose at the pose oplus(update) * Tcw: computeError + linearizeOplus
    (types_dust_tracking.cpp:64-140) -> dict(err, level, J [6], uv [2], pose64 [4, 4])."""
    dust = np.ascontiguousarray(dust, np.float32)
    X = np.ascontiguousarray(Xw, np.float32).reshape(3)
    Tin = np.ascontiguousarray(Tcw, np.float32).reshape(16)
    upd = None if update is None else np.ascontiguousarray(update, np.float64).reshape(6)
    err, level = C.c_double(0), C.c_int(0)
    J, uv, P64 = np.zeros(6, np.float64), np.zeros(2, np.float32), np.zeros(16, np.float64)
    L = lib()
    L.oracle_dust_edge.restype = None
    L.oracle_dust_edge.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                   C.c_float, C.c_float, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p, C.c_void_p,
                                   C.c_void_p]
    L.oracle_dust_edge(dust.ctypes.data, dust.shape[0], dust.shape[1], X.ctypes.data, Tin.ctypes.data,
                       None if upd is None else upd.ctypes.data, float(fx), float(fy), float(cx), float(cy),
                       C.byref(err), C.byref(level), J.ctypes.data, uv.ctypes.data, P64.ctypes.data)
    return dict(err=err.value, level=level.value, J=J, uv=uv, pose64=P64.reshape(4, 4))


def match_knn2(query, train):
    """Exact 2-nearest-neighbour search (what flann->knnMatch(query, matches, 2) approximates)."""
    q = np.ascontiguousarray(query, np.float32).reshape(-1, 256)
    t = np.ascontiguousarray(train, np.float32).reshape(-1, 256)
    idx = np.full((max(len(q), 1), 2), -1, np.int32)
    dist = np.full((max(len(q), 1), 2), np.finfo(np.float32).max, np.float32)
    L = lib()
    L.oracle_match_knn2.restype = None
    L.oracle_match_knn2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    if len(q):
        L.oracle_match_knn2(q.ctypes.data, len(q), (t if len(t) else np.zeros((1, 256), np.float32)).ctypes.data, len(t),
                            idx.ctypes.data, dist.ctypes.data)
    return idx[:len(q)], dist[:len(q)]
