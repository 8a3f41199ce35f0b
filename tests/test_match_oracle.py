"""CPU: known-answer tests of the matching oracle (oracle_match_bruteforce) against a literal
numpy statement of OpenCV's BFMatcher(NORM_L2, crossCheck) rule, and the tie / empty rules
(reference call site sp_matcher.cpp:1642-1674)."""
import numpy as np

from oracle import oracle


def _unit(rng, n):
    d = rng.standard_normal((n, 256)).astype(np.float32)
    return d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)


def _bf_numpy(q, t, cross):
    """float64 distances; valid as an index check when there are no near-ties."""
    D = np.sqrt(((q[:, None, :].astype(np.float64) - t[None, :, :]) ** 2).sum(-1))
    idx = np.full(len(q), -1, np.int32)
    if not cross:
        return D.argmin(1).astype(np.int32)
    tq = D.argmin(0)               # nearest query of every train row
    best = np.full(len(q), np.inf)
    for j in range(len(t)):
        i = tq[j]
        if D[i, j] < best[i]:
            best[i] = D[i, j]
            idx[i] = j
    return idx


def test_random_sets_agree_with_numpy_rule():
    rng = np.random.default_rng(5)
    q, t = _unit(rng, 173), _unit(rng, 140)
    t[:60] = q[20:80] + 0.05 * _unit(rng, 60)       # real correspondences
    for cross in (True, False):
        idx, dist = oracle.match_bruteforce(q, t, cross)
        assert np.array_equal(idx, _bf_numpy(q, t, cross))
        m = idx >= 0
        ref = np.linalg.norm(q[m].astype(np.float64) - t[idx[m]], axis=1)
        assert np.abs(dist[m] - ref).max() < 1e-6 and np.all(dist[~m] == np.finfo(np.float32).max)
    idx, _ = oracle.match_bruteforce(q, t, True)
    assert np.array_equal(idx[20:80], np.arange(60))   # the planted pairs are mutual


def test_cross_check_is_opencvs_vote_rule_not_mutual_nn():
    # 1-D layout embedded in 256-D: queries at 0 and 10; trains at 4 and 6.
    # Both trains vote for ... t0 (4) -> q0 (dist 4), t1 (6) -> q1 (dist 4).  Each query matched.
    def pts(xs):
        a = np.zeros((len(xs), 256), np.float32)
        a[:, 0] = xs
        return a
    idx, dist = oracle.match_bruteforce(pts([0, 10]), pts([4, 6]), True)
    assert idx.tolist() == [0, 1] and dist.tolist() == [4.0, 4.0]
    # trains at 1 and 2: both vote for q0; q0 keeps the closer (t0); q1 gets nothing.
    idx, dist = oracle.match_bruteforce(pts([0, 10]), pts([1, 2]), True)
    assert idx.tolist() == [0, -1] and dist[0] == 1.0 and dist[1] == np.finfo(np.float32).max
    # without cross-check every query gets its nearest train
    idx, _ = oracle.match_bruteforce(pts([0, 10]), pts([1, 2]), False)
    assert idx.tolist() == [0, 1]


def test_ties_go_to_the_lowest_index():
    rng = np.random.default_rng(9)
    base = _unit(rng, 4)
    q = np.stack([base[0], base[0], base[1]])          # duplicate queries 0, 1
    t = np.stack([base[0], base[0], base[1], base[1]])  # duplicate trains
    idx, dist = oracle.match_bruteforce(q, t, True)
    # trains 0,1 vote for query 0 (lowest of the tied queries); query 0 keeps train 0; query 1 unvoted;
    # trains 2,3 vote for query 2, which keeps train 2.
    assert idx.tolist() == [0, -1, 2] and dist[0] == 0.0 and dist[2] == 0.0
    idx, _ = oracle.match_bruteforce(q, t, False)
    assert idx.tolist() == [0, 0, 2]


def test_empty_and_nan():
    rng = np.random.default_rng(3)
    q = _unit(rng, 5)
    idx, dist = oracle.match_bruteforce(q, np.zeros((0, 256), np.float32), True)
    assert idx.tolist() == [-1] * 5
    idx, dist = oracle.match_bruteforce(np.zeros((0, 256), np.float32), q, True)
    assert len(idx) == 0
    t = q.copy()
    t[2, 7] = np.nan                                     # a NaN descriptor never matches
    for cross in (True, False):
        idx, _ = oracle.match_bruteforce(q, t, cross)
        assert idx.tolist() == [0, 1, -1 if cross else idx[2], 3, 4] and idx[2] != 2


def test_knn2_oracle_against_numpy_sort():
    """oracle_match_knn2 (exact 2-NN, what the reference's FLANN knnMatch(.., 2) approximates) against a literal
    numpy statement: sort the distances of every query stably, take the first two."""
    rng = np.random.default_rng(11)
    q = rng.standard_normal((40, 256)).astype(np.float32)
    t = rng.standard_normal((90, 256)).astype(np.float32)
    t[10] = t[3]
    q[0] = t[3]
    idx, dist = oracle.match_knn2(q, t)
    full = np.stack([oracle.match_bruteforce(q, t[j:j + 1], False)[1] for j in range(len(t))], 1)   # [nq, nt] distances
    order = np.argsort(full, axis=1, kind="stable")[:, :2]
    assert np.array_equal(idx, order.astype(np.int32))
    assert np.array_equal(dist, np.take_along_axis(full, order, 1))
    assert tuple(idx[0]) == (3, 10)
    i1, d1 = oracle.match_knn2(q, t[:1])
    assert (i1[:, 0] == 0).all() and (i1[:, 1] == -1).all()
