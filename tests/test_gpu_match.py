"""GPU: spfe_match / spfe_match_records_device vs the oracle (bit-exact indices AND distances).
SURVEY.md §8(f) rank 1; reference call site sp_matcher.cpp:1642-1674."""
import numpy as np
import pytest

from oracle import oracle
from sp_orb_slam_amd import synth, weights
from sp_orb_slam_amd.extractor import SPExtractor

pytestmark = pytest.mark.gpu
FMAX = np.finfo(np.float32).max


def _unit(rng, n):
    d = rng.standard_normal((n, 256)).astype(np.float32)
    return d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)


@pytest.fixture(scope="module")
def ext():
    e = SPExtractor(1000, 120, 160, weights.synthetic(7, "dense"), max_batch=4, with_heat=False)
    yield e
    e.close()


def _same(ext, q, t, cross):
    idx, dist = ext.match(q, t, cross)
    ridx, rdist = oracle.match_bruteforce(q, t, cross)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
    return idx, dist


@pytest.mark.parametrize("nq,nt", [(1, 1), (3, 200), (64, 64), (65, 63), (257, 129), (1001, 1001), (700, 2300)])
@pytest.mark.parametrize("cross", [True, False])
def test_match_random_sets(ext, nq, nt, cross):
    rng = np.random.default_rng(nq * 7 + nt)
    q, t = _unit(rng, nq), _unit(rng, nt)
    n = min(nq, nt) // 2
    t[:n] = q[nq - n:] + np.float32(0.1) * _unit(rng, n)
    idx, _ = _same(ext, q, t, cross)
    if n and cross:
        assert (idx >= 0).sum() >= n // 2


def test_match_ties_duplicates_and_distances_bits(ext):
    rng = np.random.default_rng(11)
    base = _unit(rng, 40)
    q = base[rng.integers(0, 40, 300)]                  # many exact duplicates -> many exact ties
    t = base[rng.integers(0, 40, 280)]
    for cross in (True, False):
        idx, dist = _same(ext, q, t, cross)
        assert np.all(dist[idx >= 0] == 0.0)
    # raw (un-normalised, large-range) data: the sqrt / fma chain bits
    q = (rng.standard_normal((150, 256)) * 10 ** rng.uniform(-3, 3, (150, 1))).astype(np.float32)
    t = (rng.standard_normal((170, 256)) * 10 ** rng.uniform(-3, 3, (170, 1))).astype(np.float32)
    _same(ext, q, t, True)
    _same(ext, q, t, False)


def test_match_empty_nan_inf(ext):
    rng = np.random.default_rng(2)
    q = _unit(rng, 9)
    idx, dist = ext.match(q, np.zeros((0, 256), np.float32))
    assert idx.tolist() == [-1] * 9 and np.all(dist == FMAX)
    idx, dist = ext.match(np.zeros((0, 256), np.float32), q)
    assert len(idx) == 0
    t = q.copy()
    t[2, 7] = np.nan
    t[5, 0] = np.inf
    q2 = q.copy()
    q2[7, 3] = np.nan
    for cross in (True, False):
        _same(ext, q2, t, cross)


def test_match_records_on_device(ext):
    """Two batches of frames extracted into HBM, matched there, only the match table read back."""
    import torch

    H, W, B = 120, 160, 4
    a = synth.make_batch(500, B, H, W)
    b = np.roll(a, (8, 16), axis=(1, 2))                # the same scenes shifted by whole cells: (8, 16) px
    rb = ext.record_bytes()
    ra = torch.zeros(B * rb, dtype=torch.uint8, device="cuda")
    rbb = torch.zeros(B * rb, dtype=torch.uint8, device="cuda")
    out = torch.zeros(B * ext.match_out_bytes(), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream()
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    ext.extract_batch_device(da.data_ptr(), B, ra.data_ptr(), s.cuda_stream)
    ext.extract_batch_device(db.data_ptr(), B, rbb.data_ptr(), s.cuda_stream)
    for cross in (True, False):
        ext.match_records_device(rbb.data_ptr(), ra.data_ptr(), B, out.data_ptr(), cross, s.cuda_stream)
        torch.cuda.synchronize()
        ha, hb, ho = ra.cpu().numpy(), rbb.cpu().numpy(), out.cpu().numpy()
        mb = ext.match_out_bytes()
        for i in range(B):
            fa = ext.view_record(ha[i * rb:(i + 1) * rb])
            fb = ext.view_record(hb[i * rb:(i + 1) * rb])
            idx, dist = ext.decode_match_out(ho[i * mb:(i + 1) * mb])
            ridx, rdist = oracle.match_bruteforce(fb.descriptors, fa.descriptors, cross)
            assert np.array_equal(idx[:fb.K], ridx)
            assert np.array_equal(dist[:fb.K].view(np.uint32), rdist.view(np.uint32))
            assert np.all(idx[fb.K:] == -1) and np.all(dist[fb.K:] == FMAX)
            if cross:   # the matches are geometrically the (8, 16) shift for most keypoints
                m = idx[:fb.K] >= 0
                d = fb.kp_xy[m] - fa.kp_xy[idx[:fb.K][m]]
                assert np.mean((d[:, 0] == 16) & (d[:, 1] == 8)) > 0.5


@pytest.mark.parametrize("nq,nt", [(1, 1), (5, 2), (64, 64), (65, 63), (300, 1001), (1001, 700), (2, 2300)])
def test_match_knn2_exact(ext, nq, nt):
    """spfe_match_knn2 = knnMatch(query, matches, 2), exact: indices and distances bit-identical to the oracle,
    including duplicated train rows (ties -> lower index first), a single train row (second = none) and NaNs."""
    rng = np.random.default_rng(nq * 13 + nt)
    q, t = _unit(rng, nq), _unit(rng, nt)
    if nt >= 8:
        t[5] = t[2]                       # exact duplicate: tie between train rows 2 and 5
        q[0] = t[2]
        t[7, 3] = np.nan                  # never a neighbour
    idx, dist = ext.match_knn2(q, t)
    ridx, rdist = oracle.match_knn2(q, t)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
    if nt >= 8:
        assert tuple(idx[0]) == (2, 5) and dist[0, 0] == dist[0, 1] == 0.0
        assert not (idx == 7).any()
    if nt == 1:
        assert (idx[:, 1] == -1).all() and (dist[:, 1] == FMAX).all()
    # the ratio test of KeyFrame::matchMps (keyframe.cpp:462): well defined on the exact neighbours
    good = dist[:, 0] < 0.7 * dist[:, 1]
    assert good.dtype == bool
