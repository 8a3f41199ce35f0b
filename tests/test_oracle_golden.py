"""CPU: the C oracle against the ATen-CPU golden fixtures (tests/golden/*.npz).

Integer outputs (keypoints, occ_grid, candidate count, sort order) must be
exact; float outputs within the tolerances of SURVEY.md §8c: descriptors
max-abs <= 2e-5, heat abs <= 1e-5, cov2 rel <= 1e-5, logits rel 1e-5.
"""
import os

import numpy as np
import pytest

from oracle import oracle
from sp_orb_slam_amd import synth, weights

CASES = ["g64x96_dense", "g64x96_sparse", "g128x160_sparse", "g480x752_dense", "g480x640_sparse",
         "g720x1280_sparse", "g64x376_dense_cudadiv", "g480x752_dense_cudadiv"]

DESC_TOL = 2e-5
# Fixtures made with libtorch-CUDA's form of `pixels.div(W / 2.0)` (sp_extractor.cpp:137-138: a * (1.0f / b), tools/
# aten_path.py cuda_scalar_div) — the form the oracle and the kernels pin (the reference hard-wires CUDA, :73).  End to end the
# oracle lands within the convolutions' summation-order noise of them (measured 2.0e-7 ... 2.2e-7); against the ATen-CPU form
# (true division; the fixtures above) the same frames read 5.4e-7 — the 2e-5 bar hid which form the oracle follows, this one
# does not.  Stage level (golden coarse map in, descriptors out): 6e-8.
DESC_TOL_CUDA_FORM = 3e-7
DESC_TOL_CUDA_FORM_STAGE = 1e-7
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
        assert np.abs(out["desc"] - g["kp_desc"]).max() <= (DESC_TOL_CUDA_FORM if name.endswith("_cudadiv") else DESC_TOL)
    else:
        assert np.abs(out["desc"][::16] - g["kp_desc_sub"]).max() <= (DESC_TOL_CUDA_FORM if name.endswith("_cudadiv") else DESC_TOL)
        assert np.allclose(out["semi"][::7, ::9], g["semi_sub"], rtol=1e-5, atol=2e-5)
        assert np.allclose(out["coarse"][::13, ::11], g["coarse_sub"], rtol=1e-5, atol=2e-5)
        assert np.abs(out["heat"][out["heat"].shape[0] // 2] - g["heat_row"]).max() <= HEAT_TOL


def test_sampling_coordinates_follow_the_cuda_scalar_division(golden_dir):
    """the reciprocal form of `x / (W / 2)` confirmed by a fixture instead of hidden under 2e-5.  The
    fixture's own record says how far ATen-CPU's true division is from it on this frame; the oracle's sampling stage, fed the
    fixture's coarse map and candidates, must be an order of magnitude closer than that."""
    g = np.load("%s/g64x376_dense_cudadiv.npz" % golden_dir)
    H, W = int(g["meta_H"]), int(g["meta_W"])
    assert int(g["meta_cuda_scalar_div"]) == 1
    gap = float(g["meta_desc_max_abs_diff_to_cpu_div_form"])
    assert gap >= 3e-7 and int(g["meta_desc_rows_differing_from_cpu_div_form"]) >= 5   # the two forms do differ here
    d = oracle.sample_desc(g["coarse_raw"], H, W, g["cand_xy"][:, 0].copy(), g["cand_xy"][:, 1].copy())
    err = float(np.abs(d - g["cand_desc"]).max())
    assert err <= DESC_TOL_CUDA_FORM_STAGE and err * 3 < gap
    g2 = np.load("%s/g480x752_dense_cudadiv.npz" % golden_dir)
    assert float(g2["meta_desc_max_abs_diff_to_cpu_div_form"]) >= 5e-7 and int(g2["meta_desc_rows_differing_from_cpu_div_form"]) > 1000


@pytest.mark.parametrize("name", ["g64x96_dense", "g64x96_sparse", "g128x160_sparse", "g64x376_dense_cudadiv"])
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


def test_flip_report_is_committed_and_says_what_survey_8c_expects(golden_dir):
    """tests/golden/flip_report.json (tools/flip_report.py, build container): oracle vs the ATen-CPU op sequence over 64 seeded
    frames x {640x480, 752x480, 1280x720} x {dense, sparse}.  SURVEY.md 8c: "report #cells with flipped argmax/threshold
    (expected ~0 for margin > 1e-5)" — every flipped cell's own margin must be below 1e-5, and flips must be rare."""
    import json
    rep = json.load(open("%s/flip_report.json" % golden_dir))
    assert set(rep["configs"]) == {"%s_%s" % (s, d) for s in ("640x480", "752x480", "1280x720") for d in ("dense", "sparse")}
    for key, c in rep["configs"].items():
        assert c["frames"] >= 64
        cells = c["frames"] * c["cells_per_frame"]
        assert c["arg_flips_total"] + c["thr_flips_total"] <= cells * 1e-5, key          # (measured: 7 of 3.3 M cells overall)
        assert all(g < 1e-5 for g in c["flipped_cells_top2_gaps"] + c["flipped_cells_threshold_gaps"]), key
        assert c["keypoints_on_one_side_only_total"] <= c["keypoints_total"] * 1e-4, key
        assert c["logit_max_abs_diff"] < 5e-5, key


def test_bf16_flip_report_supports_the_bf16_bounds():
    """tests/golden/flip_report_bf16.json (tools/flip_report_bf16.py: the oracle's bf16 emulation against the f32 oracle, 64
    seeded frames x {1280x720, 752x480} x {dense, sparse}) is what the bf16 tolerances of tests/test_gpu_bf16.py and of
    bench.py's parity rule stand on: SURVEY.md §8(c)'s "descriptor max-abs <= 2e-2, cos >= 0.999" hold with a wide margin, the
    descriptor of a keypoint moves by at most a few percent of the matcher's tightest threshold (0.3, sp_matcher.cpp:18), and
    the keypoint-set Jaccard over 256 frames never falls below the asserted floor."""
    import json
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "flip_report_bf16.json")))
    assert set(d["configs"]) == {"1280x720_dense", "1280x720_sparse", "752x480_dense", "752x480_sparse"}
    for name, c in d["configs"].items():
        assert c["frames"] == 64, name
        assert c["desc_max_abs_max"] <= 2e-2 and c["desc_max_abs_max"] <= 5e-3, name
        assert c["desc_cos_min"] >= 0.999 and c["desc_cos_min"] >= 0.9999, name
        assert c["desc_l2_of_common_keypoints"]["max"] <= 0.03 and c["desc_l2_rows_above"]["0.03"] == 0, name
        assert c["jaccard_min"] >= 0.87 + 0.015, name          # the GPU test's floor sits below every frame of the report
        assert 0.0 < c["arg_flips_per_cell"] < 0.05, name      # bf16 logits DO flip near-ties: ~2 % of the cells
        assert c["keypoints_common_total"] >= 0.95 * min(c["keypoints_f32_total"], c["keypoints_bf16_total"]), name


def test_bf16_flip_report_at_3840x2160():
    """The same report at the largest frame the path takes (round 6; DESIGN §12.5 of round 5): tests/golden/flip_report_bf16_2160p.json,
    4 seeded 3840x2160 frames x {dense, sparse} (129,600 cells each; 500 s of CPU) — the bf16 mode behaves there as at the sizes
    above: ~2 % arg-max flips, Jaccard 0.90 - 0.92, descriptors within 2.6 % of the matcher's tightest threshold."""
    import json
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "flip_report_bf16_2160p.json")))
    assert set(d["configs"]) == {"3840x2160_dense", "3840x2160_sparse"}
    for name, c in d["configs"].items():
        assert c["frames"] == 4 and c["cells_per_frame"] == 129600, name
        assert c["desc_max_abs_max"] <= 5e-3 and c["desc_cos_min"] >= 0.9999, name
        assert c["desc_l2_of_common_keypoints"]["max"] <= 0.03 and c["desc_l2_rows_above"]["0.03"] == 0, name
        assert c["jaccard_min"] >= 0.87 + 0.015, name
        assert 0.0 < c["arg_flips_per_cell"] < 0.05, name
        # (a Jaccard of 0.896 is 94.5 % of the keypoints in common: 3,801 / 3,830 of 4,004 here)
        assert c["keypoints_common_total"] >= 0.94 * min(c["keypoints_f32_total"], c["keypoints_bf16_total"]), name


def test_bf16_flip_report_tool_runs_on_a_small_frame():
    """The report's per-frame function on one 64x96 frame (CPU, a second): same fields, same invariants."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from tools import flip_report_bf16
    from sp_orb_slam_amd import synth, weights
    r = flip_report_bf16.one_frame(oracle, weights.synthetic(7, "dense"), synth.make_image(300, 64, 96), 50)
    assert r["cells"] == 96 and 0.0 <= r["jaccard"] <= 1.0 and r["common"] <= min(r["K_f32"], r["K_bf16"])
    if r["common"]:
        assert r["desc_cos"].min() >= 0.999 and r["desc_max_abs"] <= 2e-2
