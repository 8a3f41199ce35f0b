"""Seeded synthetic 8-bit frames (integer arithmetic only: identical everywhere).

The reference's datasets (EuRoC, Tsukuba) are not in the snapshot, so the
extractor is exercised on frames generated here: low-pass filtered noise plus
rectangles and a checkerboard patch, which give a SuperPoint-style detector
corners, edges and flat regions.
"""
import numpy as np


def _xorshift32(state, n):
    """n uint32 values from a 32-bit xorshift generator (vectorised by blocks)."""
    out = np.empty(n, np.uint32)
    s = np.uint32(state if state else 1)
    # sequential generator, but cheap: run a python loop over 64 streams
    lanes = 64
    st = (np.arange(lanes, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(s)) & np.uint64(0xFFFFFFFF)
    st = st.astype(np.uint32)
    st[st == 0] = 1
    per = (n + lanes - 1) // lanes
    buf = np.empty((per, lanes), np.uint32)
    for i in range(per):
        st ^= (st << np.uint32(13))
        st ^= (st >> np.uint32(17))
        st ^= (st << np.uint32(5))
        buf[i] = st
    out[:] = buf.reshape(-1)[:n]
    return out


def _box_blur_u32(img, r):
    """(2r+1)^2 box sum via integer cumulative sums, edge-replicated."""
    H, W = img.shape
    p = np.pad(img.astype(np.int64), r, mode="edge")
    c = np.cumsum(np.cumsum(p, 0), 1)
    c = np.pad(c, ((1, 0), (1, 0)))
    k = 2 * r + 1
    s = c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]
    return (s // (k * k)).astype(np.int64)[:H, :W]


def make_image(seed, H, W):
    """Deterministic u8 [H, W] test frame."""
    rnd = _xorshift32(seed * 7919 + 17, H * W + 64)
    noise = (rnd[:H * W] >> np.uint32(24)).astype(np.int64).reshape(H, W)
    base = _box_blur_u32(noise, 2)
    base = _box_blur_u32(base, 2)
    img = (base - 128) * 3 + 128  # stretch contrast after the blur
    # rectangles with random grey levels
    nrect = 6 + seed % 5
    for i in range(nrect):
        a, b, c, d, g = [int(v) for v in rnd[H * W + 5 * i: H * W + 5 * i + 5]]
        y0, x0 = a % max(H - 8, 1), b % max(W - 8, 1)
        hh, ww = 4 + c % max(H // 4, 1), 4 + d % max(W // 4, 1)
        img[y0:y0 + hh, x0:x0 + ww] = (img[y0:y0 + hh, x0:x0 + ww] + (g % 256)) // 2
    # checkerboard patch
    cy, cx = (seed * 37) % max(H // 2, 1), (seed * 91) % max(W // 2, 1)
    ph, pw = min(H // 3, 96), min(W // 3, 128)
    yy, xx = np.mgrid[0:ph, 0:pw]
    cb = (((yy // 12) + (xx // 12)) % 2) * 160 + 40
    img[cy:cy + ph, cx:cx + pw] = cb[:min(ph, H - cy), :min(pw, W - cx)]
    return np.clip(img, 0, 255).astype(np.uint8)


def make_batch(seed0, n, H, W):
    return np.stack([make_image(seed0 + i, H, W) for i in range(n)])
