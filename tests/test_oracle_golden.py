"""CPU: the C oracle against the ATen-CPU golden fixtures (tests/golden/*.npz).

Integer outputs (keypoints, occ_grid, candidate count, sort order) must be
exact; float outputs within the tolerances of SURVEY.md §8c: descriptors
max-abs <= 2e-5, heat abs <= 1e-5, cov2 rel <= 1e-5, logits rel 1e-5.
"""
import numpy as np
import pytest

from oracle import oracle
from sp_orb_slam_amd import synth, weights

CASES = ["g64x96_dense", "g64x96_sparse", "g128x160_sparse", "g480x752_dense", "g480x640_sparse",
         "g720x1280_sparse"]

DESC_TOL = 2e-5
HEAT_TOL = 1e-5
COV_RTOL = 1e-5


def _run(g):
    H, W = int(g["meta_H"]), int(g["meta_W"])
    img = g["image"] if "image" in g else synth.make_image(int(g["meta_image_seed"]), H, W)
    blob = weights.synthetic(int(g["meta_weight_seed"]), str(g["meta_detector"]))
    return oracle.extract(blob, img, int(g["meta_num_features"])), img


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_aten_golden(name, golden_dir):
    g = np.load("%s/%s.npz" % (golden_dir, name))
    out, _ = _run(g)
    assert out["n_candidates"] == int(g["n_candidates"])
    assert out["K"] == len(g["kp_xy"])
    assert np.array_equal(out["kp_xy"].astype(np.int16), g["kp_xy"])  # bit-exact positions
    assert np.array_equal(out["occ_grid"], g["occ_grid"])
    assert np.abs(out["response"] - g["response"]).max() <= HEAT_TOL
    assert (np.abs(out["cov2"] - g["cov2"]) / g["cov2"]).max() <= COV_RTOL
    assert (np.abs(out["cov2_inv"] - g["cov2_inv"]) / g["cov2_inv"]).max() <= COV_RTOL
    assert np.abs(out["dense_dust"] - g["dense_dust"]).max() <= 1e-5
    assert np.allclose(out["semi_dust"], g["semi_dust"], rtol=1e-5, atol=2e-5)
    if "semi" in g.files:
        assert np.allclose(out["semi"], g["semi"], rtol=1e-5, atol=2e-5)
        assert np.allclose(out["coarse"], g["coarse_raw"], rtol=1e-5, atol=2e-5)
        assert np.abs(out["heat"] - g["heat"]).max() <= HEAT_TOL
        assert np.abs(out["heat_inv"] - g["heat_inv"]).max() <= HEAT_TOL
        assert np.abs(out["desc"] - g["kp_desc"]).max() <= DESC_TOL
    else:
        assert np.abs(out["desc"][::16] - g["kp_desc_sub"]).max() <= DESC_TOL
        assert np.allclose(out["semi"][::7, ::9], g["semi_sub"], rtol=1e-5, atol=2e-5)
        assert np.allclose(out["coarse"][::13, ::11], g["coarse_sub"], rtol=1e-5, atol=2e-5)
        assert np.abs(out["heat"][out["heat"].shape[0] // 2] - g["heat_row"]).max() <= HEAT_TOL


@pytest.mark.parametrize("name", ["g64x96_dense", "g64x96_sparse", "g128x160_sparse"])
def test_oracle_stages_against_golden(name, golden_dir):
    """Stage-by-stage, feeding each oracle stage the GOLDEN upstream data."""
    g = np.load("%s/%s.npz" % (golden_dir, name))
    H, W = int(g["meta_H"]), int(g["meta_W"])
    t = oracle.tail(g["semi"], H, W)
    assert np.array_equal(np.stack([t["x"], t["y"]], 1), g["cand_xy"])  # candidates exact
    assert np.abs(t["score"] - g["cand_score"]).max() <= 1e-6
    assert np.abs(t["heat_log"] - g["heat_log"]).max() <= 2e-6
    d = oracle.sample_desc(g["coarse_raw"], H, W, g["cand_xy"][:, 0].copy(), g["cand_xy"][:, 1].copy())
    assert np.abs(d - g["cand_desc"]).max() <= DESC_TOL
    assert np.array_equal(oracle.sort(g["cand_score"]), g["order"])
    srt = g["cand_xy"][g["order"]]
    kx, ky, src, occ = oracle.nms(srt[:, 0].copy(), srt[:, 1].copy(), int(g["meta_num_features"]), W, H)
    assert np.array_equal(np.stack([kx, ky], 1).astype(np.int16), g["kp_xy"])
    assert np.array_equal(src, g["kp_src"])
    assert np.array_equal(occ, g["occ_grid"])
    h, hi, _ = oracle.heat(g["heat_log"])
    assert np.array_equal(h, g["heat"]) and np.array_equal(hi, g["heat_inv"])  # same rule -> same bits
    cov, cinv, resp = oracle.covariance(g["heat_inv"], kx, ky)
    assert np.array_equal(resp, g["response"])
    assert np.allclose(cov, g["cov2"], rtol=COV_RTOL) and np.allclose(cinv, g["cov2_inv"], rtol=COV_RTOL)
