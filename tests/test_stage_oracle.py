"""CPU: known-answer tests of the input-staging oracle (oracle_stage_input) — OpenCV 3.x remap
(INTER_LINEAR, fixed point) + crop + cvtColor, SURVEY.md §8(f) rank 2 (reference call sites:
data_loader.cc:519-521, system.cpp:160-161, mono_tracker.cpp:18-28)."""
import numpy as np

from oracle import oracle


def _identity_maps(hs, ws):
    ys, xs = np.mgrid[0:hs, 0:ws].astype(np.float32)
    return xs, ys


def test_identity_map_is_identity_and_gray_of_gray_is_gray():
    rng = np.random.default_rng(1)
    g = rng.integers(0, 256, (40, 56), dtype=np.uint8)
    bgr = np.repeat(g[:, :, None], 3, 2)           # cv::imread of a gray PNG: B = G = R
    mx, my = _identity_maps(40, 56)
    out = oracle.stage_input(bgr, 32, 48, mx, my)
    # interior: identity; the last source row/column is not an "inlier" but still reads itself
    assert np.array_equal(out, g[:32, :48])
    assert np.array_equal(oracle.stage_input(bgr, 40, 56, mx, my), g)
    assert np.array_equal(oracle.stage_input(bgr, 32, 48), g[:32, :48])   # no remap: crop + gray


def test_cvtcolor_coefficients_and_channel_order():
    px = np.zeros((8, 8, 3), np.uint8)
    px[..., 0], px[..., 1], px[..., 2] = 10, 100, 200        # B, G, R
    want_bgr = (10 * 1868 + 100 * 9617 + 200 * 4899 + 8192) >> 14
    want_rgb = (200 * 1868 + 100 * 9617 + 10 * 4899 + 8192) >> 14
    assert oracle.stage_input(px, 8, 8)[0, 0] == want_bgr == 120
    assert oracle.stage_input(px, 8, 8, rgb=True)[0, 0] == want_rgb
    px4 = np.concatenate([px, np.full((8, 8, 1), 77, np.uint8)], 2)      # alpha ignored
    assert oracle.stage_input(px4, 8, 8)[0, 0] == want_bgr
    white = np.full((8, 8, 3), 255, np.uint8)
    assert oracle.stage_input(white, 8, 8)[0, 0] == 255


def test_fixed_point_bilinear_and_rounding():
    src = np.zeros((8, 8), np.uint8)
    src[2, 3], src[2, 4], src[3, 3], src[3, 4] = 10, 50, 90, 250
    mx = np.full((8, 8), 3.5, np.float32)
    my = np.full((8, 8), 2.25, np.float32)
    # fx = 16, fy = 8 -> weights (24*16, 24*16, 8*16, 8*16) * 32
    want = (10 * 24 * 16 * 32 + 50 * 24 * 16 * 32 + 90 * 8 * 16 * 32 + 250 * 8 * 16 * 32 + 16384) >> 15
    assert oracle.stage_input(src, 8, 8, mx, my)[0, 0] == want == 65
    # 5-bit quantisation with round-half-even: 3 + 1/64 -> sx = 96.5 -> 96 (even) -> exact pixel
    mx2 = np.full((8, 8), 3 + 1 / 64, np.float32)
    my2 = np.full((8, 8), 2.0, np.float32)
    assert oracle.stage_input(src, 8, 8, mx2, my2)[0, 0] == 10
    # 3 + 3/64 -> sx = 97.5 -> 98 -> fx = 2
    mx3 = np.full((8, 8), 3 + 3 / 64, np.float32)
    assert oracle.stage_input(src, 8, 8, mx3, my2)[0, 0] == (10 * 30 * 32 * 32 + 50 * 2 * 32 * 32 + 16384) >> 15


def test_border_constant_zero():
    src = np.full((8, 8), 200, np.uint8)
    mx, my = _identity_maps(8, 8)
    out = oracle.stage_input(src, 8, 8, mx - 0.5, my)           # half a pixel to the left
    assert out[3, 0] == 100 and out[3, 1] == 200                # tap at x = -1 reads 0
    out = oracle.stage_input(src, 8, 8, mx + 0.5, my)
    assert out[3, 7] == 100 and out[3, 6] == 200                # tap at x = 8 reads 0
    out = oracle.stage_input(src, 8, 8, mx + 100.0, my)         # fully outside
    assert not out.any()
    out = oracle.stage_input(src, 8, 8, mx - 1.0, my - 1.0)
    assert out[0, 0] == 0 and out[1, 1] == 200 and out[0, 5] == 0
