"""CPU: known-answer tests of oracle_match_patches — the patch-wise association loop of
Tracker::trackFrameDustKFLocal (tracker_dust.cpp:113-172)."""
import numpy as np

from oracle import oracle


def _unit(rng, n):
    d = rng.standard_normal((n, 256)).astype(np.float32)
    return d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)


def _grid(hc, wc, cells):
    occ = np.full((hc, wc), -1, np.int16)
    for k, (v, u) in enumerate(cells):
        occ[v, u] = k
    return occ


def test_nearest_of_the_four_cells_and_threshold():
    rng = np.random.default_rng(1)
    kd = _unit(rng, 4)
    occ = _grid(6, 8, [(2, 3), (3, 3), (2, 4), (3, 4)])          # the 2 x 2 block at (u, v) = (3, 2)
    mp = kd[2:3] + np.float32(0.01) * _unit(rng, 1)              # closest to keypoint 2 = cell (v 2, u 4)
    assert oracle.match_patches(mp, [[3.7, 2.2]], occ, kd).tolist() == [2]
    far = _unit(rng, 1)                                          # ~sqrt(2) from everything: above 0.75
    assert oracle.match_patches(far, [[3.0, 2.0]], occ, kd).tolist() == [-1]
    assert oracle.match_patches(far, [[3.0, 2.0]], occ, kd, max_dist=2.0)[0] >= 0
    # a position one cell off only sees two of the four keypoints
    assert oracle.match_patches(kd[0:1], [[4.0, 2.0]], occ, kd).tolist() == [-1]   # kp 0 is at u = 3
    assert oracle.match_patches(kd[2:3], [[4.0, 2.0]], occ, kd).tolist() == [2]


def test_earlier_map_point_takes_the_keypoint():
    rng = np.random.default_rng(2)
    kd = _unit(rng, 2)
    occ = _grid(4, 4, [(1, 1), (1, 2)])
    a = kd[0:1] + np.float32(0.02) * _unit(rng, 1)               # both map points prefer keypoint 0
    b = kd[0:1] + np.float32(0.01) * _unit(rng, 1)               # b is even closer, but comes second
    mp = np.concatenate([a, b])
    out = oracle.match_patches(mp, [[1.0, 1.0], [1.0, 1.0]], occ, kd)
    assert out.tolist() == [0, -1]                                # keypoint 1 is ~1.4 away: no second choice
    kd2 = np.stack([kd[0], kd[0] + np.float32(0.3) * _unit(rng, 1)[0]])
    out = oracle.match_patches(mp, [[1.0, 1.0], [1.0, 1.0]], occ, kd2)
    assert out.tolist() == [0, 1]                                 # the loser falls back to its second best


def test_ties_take_the_first_cell_in_du_dv_order():
    rng = np.random.default_rng(3)
    d = _unit(rng, 1)
    kd = np.repeat(d, 4, 0)                                      # four identical keypoints
    # loop order (du, dv) = (0,0), (0,1), (1,0), (1,1) -> cells (v, u): (1,1), (2,1), (1,2), (2,2)
    occ = _grid(4, 4, [(2, 2), (1, 2), (2, 1), (1, 1)])           # keypoint 3 sits in the first cell visited
    assert oracle.match_patches(d, [[1.5, 1.5]], occ, kd).tolist() == [3]
    out = oracle.match_patches(np.repeat(d, 4, 0), [[1.5, 1.5]] * 4, occ, kd)
    assert out.tolist() == [3, 2, 1, 0]                           # then (2,1), (1,2), (2,2)


def test_positions_outside_the_grid():
    rng = np.random.default_rng(4)
    kd = _unit(rng, 1)
    occ = _grid(3, 3, [(2, 2)])
    assert oracle.match_patches(kd, [[-0.5, 1.0]], occ, kd).tolist() == [-1]
    assert oracle.match_patches(kd, [[3.0, 1.0]], occ, kd).tolist() == [-1]
    assert oracle.match_patches(kd, [[2.0, 2.9]], occ, kd).tolist() == [0]     # corner cell: neighbours out of range
    assert oracle.match_patches(kd, [[np.nan, 1.0]], occ, kd).tolist() == [-1]
