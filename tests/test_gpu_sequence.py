"""GPU: a synthetic camera sequence through the extractor + matcher, frame by frame — the stand-in for
BASELINE configs[4] (EuRoC MH_05 through the tracker: dataset, weights and SLAM back-end are absent,
SURVEY.md §8d C5).  What the tracker's front end does per frame is reproduced with the product's
boundary only: stage the raw frame (remap + gray), extract, match against the previous frame's
descriptors (SPMatcher::SearchByBruteForce rule), read occ_grid like Frame::GetFeaturesInArea.
Every frame is checked against the oracle; the matches are checked against the known camera motion."""
import numpy as np
import pytest

from oracle import oracle
from sp_orb_slam_amd import weights
from sp_orb_slam_amd.extractor import SPExtractor

pytestmark = pytest.mark.gpu


def _scene(rng, h, w):
    base = rng.integers(0, 256, (h // 6 + 2, w // 6 + 2)).astype(np.float32)
    up = np.kron(base, np.ones((6, 6), np.float32))[:h, :w]
    return np.clip(up + rng.normal(0, 6, (h, w)), 0, 255).astype(np.uint8)


def test_sequence_extract_match_track():
    H, W, nf, nframes = 240, 320, 400, 6
    rng = np.random.default_rng(21)
    world = _scene(rng, H + 64, W + 160)                 # the camera pans over this texture
    blob = weights.synthetic(7, "sparse")
    ext = SPExtractor(nf, H, W, blob)
    ext.set_staging(H, W, 3, False)                       # raw BGR frames, crop + gray on the GPU
    prev = None
    for k in range(nframes):
        ox, oy = 16 * k, 8 * (k % 2)                      # whole-cell pan: (16, +-8) px per frame
        gray = world[oy:oy + H, ox:ox + W]
        raw = np.repeat(gray[:, :, None], 3, 2).copy()    # cv::imread of a gray PNG
        fr = ext.extract_staged(raw)
        ref = oracle.extract(blob, oracle.stage_input(raw, H, W), nf)
        assert fr.K == ref["K"] and np.array_equal(fr.kp_xy, ref["kp_xy"])
        assert np.array_equal(fr.descriptors.view(np.uint32), ref["desc"].view(np.uint32))
        assert np.array_equal(fr.cov2_inv.view(np.uint32), ref["cov2_inv"].view(np.uint32))
        assert np.array_equal(fr.occ_grid, ref["occ_grid"])
        # Frame::GetFeaturesInArea-style lookup: one keypoint index per occupied cell
        ys, xs = np.nonzero(fr.occ_grid >= 0)
        idx = fr.occ_grid[ys, xs]
        assert np.array_equal(np.sort(idx), np.arange(fr.K))
        assert np.all(fr.kp_xy[idx, 0].astype(int) // 8 == xs) and np.all(fr.kp_xy[idx, 1].astype(int) // 8 == ys)
        if prev is not None:
            (pxy, pdesc, pox, poy) = prev
            midx, mdist = ext.match(fr.descriptors, pdesc, True)
            ridx, rdist = oracle.match_bruteforce(fr.descriptors, pdesc, True)
            assert np.array_equal(midx, ridx) and np.array_equal(mdist.view(np.uint32), rdist.view(np.uint32))
            m = midx >= 0
            # a matched keypoint moved by the camera motion between the two frames
            d = fr.kp_xy[m] - pxy[midx[m]]
            good = (d[:, 0] == -(ox - pox)) & (d[:, 1] == -(oy - poy))
            assert m.sum() >= 0.3 * fr.K and good.mean() > 0.8, (m.sum(), fr.K, good.mean())
            assert np.all(mdist[m][good] < 0.7)           # below SPMatcher's thresholds (sp_matcher.cpp:18-19)
        prev = (fr.kp_xy.copy(), fr.descriptors.copy(), ox, oy)
    ext.close()
