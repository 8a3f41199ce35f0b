"""A synthetic camera sequence for the tracker's front-end chain (tests, bench): a fronto-parallel textured plane at
depth z0, a camera that translates parallel to it (whole-cell pans, so the network's logits repeat from frame to
frame), map points back-projected from the previous frame's keypoints.  numpy only; no oracle, no kernels."""
import numpy as np

FX, FY, CX, CY = 458.654, 457.296, 367.215, 248.375   # EuRoC cam0 (the reference's config)
Z0 = 4.0


def texture(seed, h, w):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h // 6 + 2, w // 6 + 2)).astype(np.float32)
    up = np.kron(base, np.ones((6, 6), np.float32))[:h, :w]
    return np.clip(up + rng.normal(0, 6, (h, w)), 0, 255).astype(np.uint8)


def offsets(k, period=8):
    """Pan of frame k in pixels: a triangle wave of whole cells in x, +-1 cell in y."""
    j = k % (2 * period)
    return 16 * (j if j < period else 2 * period - j), 8 * (k % 2)


def world_size(H, W, period=8):
    return H + 16, W + 16 * period


def frame(world, k, H, W):
    ox, oy = offsets(k)
    return np.ascontiguousarray(world[oy:oy + H, ox:ox + W])


def pose(ox, oy):
    """Tcw of the camera whose image is the plane panned by (ox, oy) pixels."""
    T = np.eye(4, dtype=np.float32)
    T[0, 3] = -ox * Z0 / FX
    T[1, 3] = -oy * Z0 / FY
    return T


def map_points(kp_xy, desc, k_prev, max_points=180):
    """Map points for tracking frame k_prev + 1: up to max_points keypoints of frame k_prev (evenly spread over its raster
    order), back-projected onto the plane; their track descriptors are the keypoints' descriptors."""
    n = len(kp_xy)
    sel = np.arange(0, n, max(1, n // max_points))[:max_points]
    T = pose(*offsets(k_prev))
    Xc = np.stack([(kp_xy[sel, 0] - CX) / FX * Z0, (kp_xy[sel, 1] - CY) / FY * Z0, np.full(len(sel), Z0)], 1)
    return (Xc - T[:3, 3].astype(np.float64)).astype(np.float32), np.ascontiguousarray(desc[sel], np.float32), sel


def start_pose(k):
    """The tracker's prediction for frame k (mVelocity * mLastFrame.mTcw, tracker_dust.cpp:23): x from the motion model,
    y from the last frame — one cell off."""
    return pose(offsets(k)[0], offsets(k - 1)[1])
