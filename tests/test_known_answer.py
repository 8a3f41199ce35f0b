"""CPU: hand-made known-answer cases for the host-glue stages of the oracle
(nms: sp_extractor.cpp:161-250, computeCovariance: :252-340), independent of the
network.  Expected values are derived by hand in the comments."""
import numpy as np

from oracle import oracle

W, H = 96, 64
f32 = np.float32


def _nms(cands, nf):
    """cands: list of (x, y, score) in ANY order -> sort like :489-498, then nms."""
    xs = np.array([c[0] for c in cands], f32)
    ys = np.array([c[1] for c in cands], f32)
    sc = np.array([c[2] for c in cands], f32)
    order = oracle.sort(sc)
    kx, ky, src, occ = oracle.nms(xs[order], ys[order], nf, W, H)
    return [(int(a), int(b)) for a, b in zip(kx, ky)], occ


def test_border_reject_at_7_8_and_w_minus_9_8():
    # :222-224: keep iff 8 <= x < W-8 and 8 <= y < H-8
    kps, occ = _nms([(7, 20, 0.9), (8, 30, 0.8), (W - 9, 20, 0.7), (W - 8, 30, 0.6),
                     (40, 7, 0.5), (60, 8, 0.4), (50, H - 9, 0.3), (70, H - 8, 0.2)], 100)
    assert kps == [(60, 8), (W - 9, 20), (8, 30), (50, H - 9)]          # raster order (y, then x)
    assert occ[1, 7] == 0 and occ[2, (W - 9) // 8] == 1 and occ[3, 1] == 2 and occ[(H - 9) // 8, 6] == 3
    assert (occ >= 0).sum() == 4
    # a border candidate is rejected from the output but still suppresses its neighbours (:195-214
    # runs over all candidates, the border test comes later :222-224)
    kps, _ = _nms([(40, 7, 0.5), (40, 8, 0.4)], 100)
    assert kps == []


def test_distance_4_suppresses_distance_5_does_not():
    # :202-208 zeroes the 9x9 window: Chebyshev distance <= 4
    kps, _ = _nms([(20, 20, 0.9), (24, 20, 0.8)], 100)
    assert kps == [(20, 20)]
    kps, _ = _nms([(20, 20, 0.9), (25, 20, 0.8)], 100)
    assert kps == [(20, 20), (25, 20)]
    kps, _ = _nms([(20, 20, 0.9), (24, 24, 0.8)], 100)   # diagonal, still inside the square window
    assert kps == [(20, 20)]
    kps, _ = _nms([(20, 20, 0.8), (24, 24, 0.9)], 100)   # the better one wins wherever it is
    assert kps == [(24, 24)]


def test_suppressed_candidates_suppress_nobody():
    # A kills B; B is dead when reached (:199-200), so C (4 px from B, 8 from A) survives
    kps, _ = _nms([(20, 20, 0.9), (24, 20, 0.8), (28, 20, 0.7)], 100)
    assert kps == [(20, 20), (28, 20)]


def test_cut_counts_survivors_before_border_reject():
    # :211-213: stop after the (num_features+1)-th survivor.  nf = 2 -> 3 survivors kept in
    # score order: (4,30) [border, still counts], (60,40), (30,20); (70,50) is never reached.
    kps, occ = _nms([(30, 20, 0.5), (60, 40, 0.6), (4, 30, 0.9), (70, 50, 0.4)], 2)
    assert kps == [(30, 20), (60, 40)]
    assert (occ >= 0).sum() == 2


def test_score_ties_lower_index_first():
    # the build's tie rule: equal score -> lower candidate index ranks first
    kps, _ = _nms([(20, 20, 0.5), (23, 20, 0.5)], 100)
    assert kps == [(20, 20)]
    kps, _ = _nms([(23, 20, 0.5), (20, 20, 0.5)], 100)
    assert kps == [(23, 20)]


def _heat(points):
    h = np.zeros((H, W), f32)
    for (x, y), v in points.items():
        h[y, x] = v
    return h


def _moments(seq, x0, y0):
    """seq: [(x, y, value)] in pop order -> (cov_x, cov_y) with f32 arithmetic (:316-333)."""
    s = f32(0)
    for _, _, v in seq:
        s = f32(s + f32(v))
    cx = cy = f32(0)
    for x, y, v in seq:
        wgt = f32(f32(v) / s)
        cx = f32(cx + f32(wgt * f32((x - x0) ** 2)))
        cy = f32(cy + f32(wgt * f32((y - y0) ** 2)))
    return max(cx, f32(1)), max(cy, f32(1))


def test_covariance_single_ridge():
    # ridge along +x from the keypoint: 1.0 .9 .8 .7 .6 .5 .4 ; all else 0 (stops the walk)
    vals = [1.0, .9, .8, .7, .6, .5, .4]
    h = _heat({(10 + i, 10): v for i, v in enumerate(vals)})
    cov, cinv, resp = oracle.covariance(h, np.array([10], f32), np.array([10], f32))
    ex = _moments([(10 + i, 10, v) for i, v in enumerate(vals)], 10, 10)
    assert resp[0] == f32(1.0)
    assert cov[0, 0] == ex[0] and cov[0, 1] == f32(1.0)          # cov_y = 0 -> clamped to 1
    assert abs(float(cov[0, 0]) - 46.9 / 4.9) < 1e-5
    assert cinv[0, 0] == f32(1) / ex[0]


def test_covariance_shared_visited_mask():
    # two peaks with one valley pixel (13,10) between them; the first keypoint in raster
    # order takes the valley, the second one must not count it again (:285,295)
    row = {10: 1.0, 11: .9, 12: .8, 13: .7, 14: .8, 15: .9, 16: 1.0}
    h = _heat({(x, 10): v for x, v in row.items()})
    cov, _, _ = oracle.covariance(h, np.array([10, 16], f32), np.array([10, 10], f32))
    a = _moments([(10, 10, 1.0), (11, 10, .9), (12, 10, .8), (13, 10, .7)], 10, 10)
    b = _moments([(16, 10, 1.0), (15, 10, .9), (14, 10, .8)], 16, 10)
    assert cov[0, 0] == a[0] and cov[1, 0] == b[0]
    lone_b = _moments([(16, 10, 1.0), (15, 10, .9), (14, 10, .8), (13, 10, .7)], 16, 10)
    assert cov[1, 0] != lone_b[0]


def test_covariance_column_zero_excluded():
    # bounds are xx > 0 / yy > 0 (:303,306): column 0 is never entered
    h = _heat({(3, 10): 1.0, (2, 10): .9, (1, 10): .8, (0, 10): .7})
    cov, _, _ = oracle.covariance(h, np.array([3], f32), np.array([10], f32))
    ex = _moments([(3, 10, 1.0), (2, 10, .9), (1, 10, .8)], 3, 10)
    assert cov[0, 0] == ex[0]


def test_covariance_duplicate_pops():
    # visited is set at POP (:285), so (11,11), pushed by (11,10) and again by (10,11)
    # before its first pop, is accumulated twice.  Pop order derived by hand:
    #   c, R(11,10), D(10,11), (12,10), (11,11), (11,11), (10,12), (13,10), (14,10)
    pts = {(10, 10): 1.0, (11, 10): .9, (12, 10): .8, (13, 10): .6, (14, 10): .5,
           (10, 11): .9, (10, 12): .8, (11, 11): .7}
    h = _heat(pts)
    cov, _, _ = oracle.covariance(h, np.array([10], f32), np.array([10], f32))
    seq = [(10, 10, 1.0), (11, 10, .9), (10, 11, .9), (12, 10, .8), (11, 11, .7), (11, 11, .7),
           (10, 12, .8), (13, 10, .6), (14, 10, .5)]
    ex = _moments(seq, 10, 10)
    assert cov[0, 0] == ex[0] and cov[0, 1] == ex[1]
    nodup = _moments(seq[:5] + seq[6:], 10, 10)
    assert cov[0, 0] != nodup[0]


def test_covariance_start_pixel_always_processed():
    # the second keypoint's start pixel was already consumed by the first one's walk;
    # it is still popped (q.push(kp.pt) :275) and gives response and a 1-pixel region
    row = {10: 1.0, 11: .9, 12: .8}
    h = _heat({(x, 10): v for x, v in row.items()})
    cov, _, resp = oracle.covariance(h, np.array([10, 12], f32), np.array([10, 10], f32))
    assert resp[1] == f32(.8) and cov[1, 0] == f32(1) and cov[1, 1] == f32(1)


def test_zero_and_one_candidate_shapes():
    kx, ky, src, occ = oracle.nms(np.zeros(0, f32), np.zeros(0, f32), 10, W, H)
    assert len(kx) == 0 and (occ == -1).all()
    kx, ky, src, occ = oracle.nms(np.array([40], f32), np.array([32], f32), 10, W, H)
    assert list(kx) == [40] and occ[4, 5] == 0
    d = oracle.sample_desc(np.ones((H // 8, W // 8, 256), f32), H, W, np.zeros(0, f32), np.zeros(0, f32))
    assert d.shape == (0, 256)
