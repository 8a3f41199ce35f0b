"""GPU: the tail / selection / descriptor / covariance kernels driven with
hand-made head outputs through spfe_postprocess, against the oracle's
postprocess on the SAME semi/coarse arrays.  With identical inputs every output
must be identical: integers exactly, floats BITWISE (same operation order,
include/spfe_exact_math.h)."""
import os

import numpy as np
import pytest

from oracle import oracle
from sp_orb_slam_amd.extractor import SPExtractor
from sp_orb_slam_amd import synth, weights

pytestmark = pytest.mark.gpu
f32 = np.float32
_BLOB = None


def _blob():
    global _BLOB
    if _BLOB is None:
        _BLOB = weights.synthetic(7, "dense")
    return _BLOB


def _bits(a):
    return np.ascontiguousarray(a, f32).view(np.uint32)


def _check_exact(fr, ref, heat=True):
    assert fr.status == 0
    assert fr.n_candidates == ref["n_candidates"] and fr.K == ref["K"]
    assert np.array_equal(fr.kp_xy, ref["kp_xy"])
    assert np.array_equal(fr.occ_grid, ref["occ_grid"])
    assert np.array_equal(_bits(fr.descriptors), _bits(ref["desc"]))
    assert np.array_equal(_bits(fr.response), _bits(ref["response"]))
    assert np.array_equal(_bits(fr.cov2), _bits(ref["cov2"]))
    assert np.array_equal(_bits(fr.cov2_inv), _bits(ref["cov2_inv"]))
    assert np.array_equal(_bits(fr.dense_dust), _bits(ref["dense_dust"]))
    assert np.array_equal(_bits(fr.semi_dust), _bits(ref["semi_dust"]))
    if heat:
        assert np.array_equal(_bits(fr.heat), _bits(ref["heat"]))
        assert np.array_equal(_bits(fr.heat_inv), _bits(ref["heat_inv"]))


def _run(semi, coarse, H, W, nf):
    ext = SPExtractor(nf, H, W, _blob())
    fr = ext.postprocess(semi, coarse)[0]
    ext.close()
    return fr, oracle.postprocess(semi, coarse, H, W, nf)


def _semi_from_candidates(H, W, cands, empty_dust=20.0):
    """cands: (x, y, logit).  p = e^a / (e^a + 64) for the chosen channel."""
    hc, wc = H // 8, W // 8
    semi = np.zeros((hc, wc, 65), f32)
    semi[:, :, 64] = empty_dust
    for x, y, a in cands:
        cy, cx = y // 8, x // 8
        semi[cy, cx, :] = 0
        semi[cy, cx, (y % 8) * 8 + x % 8] = a
    return semi


def _coarse(H, W, seed=0):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((H // 8, W // 8, 256)).astype(f32)


def test_handmade_nms_cases():
    H, W = 64, 96
    cases = [
        ([(7, 20, 5.0), (8, 30, 4.8), (W - 9, 20, 4.6), (W - 8, 30, 4.4), (40, 7, 4.2), (60, 8, 4.0),
          (50, H - 9, 3.8), (70, H - 8, 3.6)], 100, [(60, 8), (W - 9, 20), (8, 30), (50, H - 9)]),
        ([(20, 20, 5.0), (24, 20, 4.0)], 100, [(20, 20)]),               # distance 4 suppresses
        ([(20, 20, 5.0), (25, 20, 4.0)], 100, [(20, 20), (25, 20)]),     # distance 5 does not
        ([(23, 23, 4.0), (24, 24, 5.0)], 100, [(24, 24)]),               # across a cell corner
        ([(20, 20, 5.0), (24, 20, 4.5), (28, 20, 4.0)], 100, [(20, 20), (28, 20)]),  # dead suppress nobody
        ([(30, 20, 4.5), (60, 40, 4.6), (4, 30, 5.0), (70, 50, 4.4)], 2, [(30, 20), (60, 40)]),  # cut before border
        ([(20, 20, 4.0), (24, 20, 4.0)], 100, [(20, 20)]),               # tie -> lower cell index
        ([(40, 7, 5.0), (40, 8, 4.0)], 100, []),                         # border candidate still suppresses
    ]
    for cands, nf, expect in cases:
        semi = _semi_from_candidates(H, W, cands)
        fr, ref = _run(semi, _coarse(H, W), H, W, nf)
        _check_exact(fr, ref)
        assert [(int(x), int(y)) for x, y in fr.kp_xy] == expect


def test_threshold_boundary_and_argmax_ties():
    H, W = 64, 96
    hc, wc = H // 8, W // 8
    # p = e^a/(e^a+64) around 0.007: a = ln(64*0.007/0.993) = -0.7959...
    a0 = np.log(64 * 0.007 / 0.993)
    semi = np.zeros((hc, wc, 65), f32)
    for i, cx in enumerate(range(1, wc - 1)):
        semi[3, cx, 20] = f32(a0 + (i - 5) * 2e-4)     # straddles the >= 0.007 compare
    semi[5, 4, 9] = 1.0
    semi[5, 4, 33] = 1.0                                # exact tie inside a cell -> lowest channel
    semi[5, 7, :64] = 0.25                              # all 64 equal -> channel 0
    fr, ref = _run(semi, _coarse(H, W, 1), H, W, 500)
    _check_exact(fr, ref)
    assert 0 < ref["n_candidates"]
    kp = {(int(x), int(y)) for x, y in fr.kp_xy}
    assert (4 * 8 + 1, 5 * 8 + 1) in kp and (7 * 8, 5 * 8) in kp


@pytest.mark.parametrize("H,W,nf,scale,seed", [(64, 96, 30, 1.0, 0), (120, 160, 1000, 2.0, 1),
                                               (240, 320, 100, 0.5, 2), (480, 752, 1000, 1.5, 3),
                                               (480, 752, 200, 3.0, 4)])
def test_random_logits_exact(H, W, nf, scale, seed):
    rng = np.random.default_rng(seed)
    semi = (rng.standard_normal((H // 8, W // 8, 65)) * scale).astype(f32)
    fr, ref = _run(semi, _coarse(H, W, seed), H, W, nf)
    _check_exact(fr, ref)
    assert fr.K > 0


@pytest.mark.parametrize("nf,levels", [(10, (4.0,)), (37, (4.0,)), (25, (5.0, 4.0)), (60, (5.0, 4.0, 4.0, 3.0))])
def test_cut_through_a_group_of_equal_scores(nf, levels):
    """The num_features+1 cut lands inside a run of identical scores (one score for the whole frame: the radix
    select's range is zero): the lowest cell indices of the run survive (sp_extractor.cpp:489-498 with the
    oracle's tie rule), on the GPU through the tie list of select_kernel."""
    H, W = 128, 160
    cands = []
    i = 0
    for y in range(12, H - 12, 10):          # >= 10 px apart: nobody suppresses anybody
        for x in range(12, W - 12, 10):
            cands.append((x, y, levels[i % len(levels)]))
            i += 1
    assert len(cands) > nf + 1
    semi = _semi_from_candidates(H, W, cands)
    fr, ref = _run(semi, _coarse(H, W, 5), H, W, nf)
    _check_exact(fr, ref)
    assert fr.K == nf + 1 or fr.K == nf    # (the nf+1-th survivor is kept, :211-213)


def test_selection_with_more_cells_than_the_register_path_holds():
    """1024 x 1088: 17,408 cells > 16 per thread — select_kernel's cut walks its keys in LDS instead of registers."""
    H, W, nf = 1024, 1088, 1000
    rng = np.random.default_rng(11)
    semi = (rng.standard_normal((H // 8, W // 8, 65)) * 1.5).astype(f32)
    fr, ref = _run(semi, _coarse(H, W, 11), H, W, nf)
    _check_exact(fr, ref)
    assert fr.K > nf // 2


def _hills(H, W, seed, sigma, nh, rough=0.35):
    """Smooth per-pixel logit field (sum of Gaussian bumps): broad hills whose BFS regions overlap."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    g = np.zeros((H, W))
    for _ in range(nh):
        cx, cy = rng.uniform(0, W), rng.uniform(0, H)
        g += rng.uniform(1, 4) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sigma ** 2))
    g += rng.standard_normal((H, W)) * rough   # roughness keeps the regions tens of pixels, not thousands
    hc, wc = H // 8, W // 8
    semi = np.zeros((hc, wc, 65), f32)
    semi[:, :, :64] = g.reshape(hc, 8, wc, 8).transpose(0, 2, 1, 3).reshape(hc, wc, 64).astype(f32)
    semi[:, :, 64] = f32(g.max() * 0.6)
    return semi


@pytest.mark.parametrize("H,W,sigma,nh,seed", [(64, 96, 6.0, 6, 0), (128, 160, 10.0, 12, 1),
                                               (240, 320, 14.0, 40, 2), (240, 320, 4.0, 200, 3)])
def test_overlapping_covariance_regions_exact(H, W, sigma, nh, seed):
    """Broad hills: many keypoints whose downhill regions collide, exercising the
    conflict-resolution rounds of the covariance kernel against the sequential loop."""
    semi = _hills(H, W, seed, sigma, nh, rough=0.7 if seed == 3 else 0.35)
    fr, ref = _run(semi, _coarse(H, W, seed), H, W, 1000)
    _check_exact(fr, ref)
    # the case is only meaningful if regions really interact: the lone walk of at least one
    # keypoint differs from the sequential result
    hinv = ref["heat_inv"]
    lone = np.array([oracle.covariance(hinv, ref["kp_xy"][i:i + 1, 0].copy(), ref["kp_xy"][i:i + 1, 1].copy())[0][0]
                     for i in range(ref["K"])])
    assert (lone != ref["cov2"]).any(), "test input has no interacting regions"


def test_covariance_queue_overflow_is_handled_on_the_device(monkeypatch):
    """Walks that outgrow the regular per-keypoint FIFO (forced here: SPFE_COV_CAPS = "qcap,ovf_slots,ovf_cap,fallback_cap,ecap" with qcap = 24) redo themselves in an
    overflow slot ON THE DEVICE: the record is complete and exact, status stays 0, no host fallback involved."""
    H, W = 128, 160
    semi = _hills(H, W, 1, 10.0, 12)
    coarse = _coarse(H, W, 1)
    monkeypatch.setenv("SPFE_COV_CAPS", "24,1024")
    ext = SPExtractor(1000, H, W, _blob())
    fr = ext.postprocess(semi, coarse)[0]
    ext.close()
    ref = oracle.postprocess(semi, coarse, H, W, 1000)
    assert fr.status == 0
    assert np.array_equal(_bits(fr.cov2), _bits(ref["cov2"])) and np.array_equal(_bits(fr.cov2_inv), _bits(ref["cov2_inv"]))
    assert np.array_equal(fr.kp_xy, ref["kp_xy"])


def test_covariance_overflow_beyond_the_slots_is_redone_on_the_device(monkeypatch):
    """No overflow slots (forced: none configured) / walks that outgrow them: the frame is redone by the device-side last
    resort (cov_fallback_kernel: the reference's sequential loop, literally, on global memory) — exact, status 0, for the
    host call AND for the device-resident record (what an all-gather would ship).  No host routine exists any more."""
    import torch
    H, W = 128, 160
    semi = _hills(H, W, 1, 10.0, 12)
    coarse = _coarse(H, W, 1)
    ref = oracle.postprocess(semi, coarse, H, W, 1000)
    for slots, cap in (("0", "16384"), ("2", "40")):
        monkeypatch.setenv("SPFE_COV_CAPS", "24,%s,%s" % (slots, cap))
        ext = SPExtractor(1000, H, W, _blob())
        fr = ext.postprocess(semi, coarse)[0]
        assert fr.status == 0
        assert np.array_equal(_bits(fr.cov2), _bits(ref["cov2"])) and np.array_equal(_bits(fr.cov2_inv), _bits(ref["cov2_inv"]))
        assert np.array_equal(fr.kp_xy, ref["kp_xy"])
        ext.close()


def test_covariance_last_resort_capacity_is_the_only_reported_failure(monkeypatch):
    """The last resort's list (SPFE_COV_CAPS field 4) forced tiny: a region with more pops than the last-resort list holds is the one case that
    leaves SPFE_STATUS_COV_OVERFLOW in the record (reported, not guessed); keypoints and descriptors are unaffected."""
    H, W = 128, 160
    semi = _hills(H, W, 1, 10.0, 12)
    coarse = _coarse(H, W, 1)
    monkeypatch.setenv("SPFE_COV_CAPS", "24,0,,1024")
    ext = SPExtractor(1000, H, W, _blob())
    fr = ext.postprocess(semi, coarse)[0]
    ext.close()
    ref = oracle.postprocess(semi, coarse, H, W, 1000)
    assert np.array_equal(fr.kp_xy, ref["kp_xy"]) and np.array_equal(_bits(fr.descriptors), _bits(ref["desc"]))
    if fr.status == 0:     # every region fits in 1024 pops on this input: then the values must be exact
        assert np.array_equal(_bits(fr.cov2), _bits(ref["cov2"]))
    else:
        assert fr.status == 1


def test_covariance_overflow_on_the_pipelined_host_path_without_heat_maps(monkeypatch):
    """spfe_submit_batch / spfe_collect_batch on a handle WITHOUT SPFE_FLAG_HEAT used to hand out invalid
    covariances when a frame overflowed (the host repair needed heat_inv, which that path never copied).  With the
    device-side last resort there is nothing to repair: tiny lists, no slots -> exact values, status 0, three batches in
    flight."""
    from sp_orb_slam_amd import synth, weights
    H, W, nf, B = 120, 160, 150, 2
    blob = weights.synthetic(7, "dense")
    monkeypatch.setenv("SPFE_COV_CAPS", "16,1,32")
    batches = [[synth.make_image(900 + 10 * s + i, H, W) for i in range(B)] for s in range(4)]
    ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False)
    tk = [ext.submit_batch(b) for b in batches[:3]]
    got = []
    for k in range(4):
        got.append(ext.collect_batch(tk.pop(0)))
        if k == 0:
            tk.append(ext.submit_batch(batches[3]))
    ext.close()
    npop_max = 0
    for b, res in zip(batches, got):
        for img, fr in zip(b, res):
            ref = oracle.extract(blob, img, nf)
            assert fr.status == 0 and fr.K == ref["K"] and np.array_equal(fr.kp_xy, ref["kp_xy"])
            assert np.array_equal(_bits(fr.cov2), _bits(ref["cov2"])) and np.array_equal(_bits(fr.cov2_inv), _bits(ref["cov2_inv"]))
            npop_max = max(npop_max, int(np.max(ref["cov2"])))
    assert npop_max > 1       # regions are wider than a pixel: the 16-entry lists did overflow


def test_covariance_wide_hills_leave_the_staged_window(monkeypatch):
    """Regions wider than the 32x32 window staged in LDS: the walk continues on global lookups and remembers the
    pixels it popped out there; a hill with more than 256 distinct pixels outside the window sends the frame to the
    device-side last resort.  Exact against the sequential oracle either way, status 0."""
    H, W = 128, 160
    for sigma, nh, rough in ((30.0, 4, 0.01), (24.0, 5, 0.03), (30.0, 4, 0.05)):
        semi = _hills(H, W, 1, sigma, nh, rough=rough)
        coarse = _coarse(H, W, 1)
        ext = SPExtractor(1000, H, W, _blob())
        fr = ext.postprocess(semi, coarse)[0]
        ext.close()
        ref = oracle.postprocess(semi, coarse, H, W, 1000)
        assert np.array_equal(fr.kp_xy, ref["kp_xy"])
        assert fr.status == 0
        assert np.array_equal(_bits(fr.cov2), _bits(ref["cov2"])) and np.array_equal(_bits(fr.cov2_inv), _bits(ref["cov2_inv"]))


def test_postprocess_batch():
    H, W, n = 64, 96, 3
    rng = np.random.default_rng(5)
    semi = (rng.standard_normal((n, H // 8, W // 8, 65)) * 1.5).astype(f32)
    coarse = rng.standard_normal((n, H // 8, W // 8, 256)).astype(f32)
    ext = SPExtractor(40, H, W, _blob(), max_batch=4)
    frs = ext.postprocess(semi, coarse)
    ext.close()
    for i in range(n):
        _check_exact(frs[i], oracle.postprocess(semi[i], coarse[i], H, W, 40))


@pytest.mark.parametrize("H,W,nf,scale,seed", [(1080, 1920, 1000, 1.5, 0), (1080, 1920, 10000, 0.2, 1), (1440, 2560, 2000, 1.0, 2),
                                               (1088, 1920, 1000, 0.0, 3)])
def test_selection_beyond_16k_cells(H, W, nf, scale, seed):
    """Frames of more than 16,384 cells (1920x1080 = 32,400; 2560x1440 = 57,600): select_kernel keeps only the cell states
    and neighbour masks in LDS and its thread-private per-cell data in global scratch (tail_select.hip, BIG) — the reference
    takes any multiple of 8 (sp_extractor.cpp:70).  Exact against the literal oracle: candidates, the cut (scale 0.2: the
    softmax scores crowd together -> many keys in the threshold bucket; scale 0: ALL scores equal -> the cut is decided by
    the index tie rule alone), border, raster order, occ_grid, and everything behind the selection."""
    rng = np.random.default_rng(seed)
    semi = (rng.standard_normal((H // 8, W // 8, 65)) * scale).astype(f32)
    coarse = rng.standard_normal((H // 8, W // 8, 256)).astype(f32)
    fr, ref = _run(semi, coarse, H, W, nf)
    _check_exact(fr, ref)
    assert fr.K > 0.5 * min(nf, ref["n_candidates"] // 9)


def test_full_extraction_1080p_matches_oracle():
    """1920x1080 through the whole f32 path (network, tail, BIG selection, descriptors, covariance): bitwise vs the oracle."""
    from sp_orb_slam_amd import synth
    H, W, nf = 1080, 1920, 1000
    blob = weights.synthetic(7, "sparse")
    img = synth.make_image(77, H, W)
    ext = SPExtractor(nf, H, W, blob, with_heat=False)
    kps, desc = ext(img, None)
    fr = ext.last
    ext.close()
    ref = oracle.extract(blob, img, nf)
    assert fr.status == 0 and fr.K == ref["K"] and np.array_equal(fr.kp_xy, ref["kp_xy"])
    assert np.array_equal(fr.occ_grid, ref["occ_grid"])
    assert np.array_equal(_bits(fr.descriptors), _bits(ref["desc"]))
    assert np.array_equal(_bits(fr.cov2_inv), _bits(ref["cov2_inv"]))


def test_frame_size_limits():
    """Up to 262,143 cells and 2^31 bytes of first-layer activations per frame: f32 3840x2160 fits, 4096x2304 (2.4 GB) does not;
    bf16 (half the bytes) ends at the cell limit (4096x4096 = 262,144 cells).  Refused at spfe_create with a message."""
    with pytest.raises(Exception, match="too large"):
        SPExtractor(1000, 2304, 4096, _blob())
    with pytest.raises(Exception, match="too large"):
        SPExtractor(1000, 4096, 4096, _blob(), precision="bf16")


@pytest.mark.parametrize("H,W,nf,scale,seed", [(2160, 3840, 1000, 1.5, 0), (2160, 3840, 10000, 0.2, 1), (2160, 3840, 3000, 0.0, 2)])
def test_selection_beyond_65535_cells(H, W, nf, scale, seed):
    """3840x2160 = 129,600 cells: select_huge_kernel (tail_select.hip — everything per cell in global scratch, 32-bit cell
    indices, 256 cells per thread) against the literal oracle on random logits: the cut among crowded scores (scale 0.2), by
    the index tie rule alone (scale 0: all scores equal), border, raster order, occ_grid, descriptors, covariance."""
    rng = np.random.default_rng(seed)
    semi = (rng.standard_normal((H // 8, W // 8, 65)) * scale).astype(f32)
    coarse = rng.standard_normal((H // 8, W // 8, 256)).astype(f32)
    fr, ref = _run(semi, coarse, H, W, nf)
    _check_exact(fr, ref)
    assert fr.K > 0.5 * min(nf, ref["n_candidates"] // 9)


def test_full_extraction_2160p_matches_oracle():
    """3840x2160 through the whole f32 path (the reference accepts any multiple of 8, sp_extractor.cpp:70): conv1a's output is
    2,123,366,400 bytes a frame — the largest the convolutions' 32-bit buffer offsets address — and the selection runs as
    select_huge_kernel.  Bitwise vs the oracle."""
    from sp_orb_slam_amd import synth
    H, W, nf = 2160, 3840, 1000
    blob = weights.synthetic(7, "sparse")
    img = synth.make_image(78, H, W)
    ext = SPExtractor(nf, H, W, blob, with_heat=False)
    kps, desc = ext(img, None)
    fr = ext.last
    assert ext.debug_read("select_huge")[0] == 1
    ext.close()
    ref = oracle.extract(blob, img, nf)
    assert fr.status == 0 and fr.K == ref["K"] and np.array_equal(fr.kp_xy, ref["kp_xy"])
    assert np.array_equal(fr.occ_grid, ref["occ_grid"])
    assert np.array_equal(_bits(fr.descriptors), _bits(ref["desc"]))
    assert np.array_equal(_bits(fr.cov2_inv), _bits(ref["cov2_inv"]))


def test_select_huge_form_on_frames_of_any_size(monkeypatch):
    """SPFE_SELECT_HUGE=1 runs the selection of every frame on select_huge_kernel: the crafted cases of this file (border,
    distance 4 / 5, cell corners, the cut before the border reject, ties by cell index, the threshold compare, cuts through
    groups of equal scores), random logits at five sizes and 1920x1080 / 2560x1440 — all exact against the literal oracle —
    and a batch of whole extractions equal to the default form's records."""
    monkeypatch.setenv("SPFE_SELECT_HUGE", "1")
    ext = SPExtractor(50, 64, 96, _blob())
    ext.postprocess(np.zeros((8, 12, 65), f32), _coarse(64, 96))
    assert ext.debug_read("select_huge")[0] == 1
    ext.close()
    test_handmade_nms_cases()
    test_threshold_boundary_and_argmax_ties()
    for args in [(64, 96, 30, 1.0, 0), (120, 160, 1000, 2.0, 1), (240, 320, 100, 0.5, 2), (480, 752, 1000, 1.5, 3), (480, 752, 200, 3.0, 4)]:
        test_random_logits_exact(*args)
    for nf, levels in [(10, (4.0,)), (37, (4.0,)), (25, (5.0, 4.0)), (60, (5.0, 4.0, 4.0, 3.0))]:
        test_cut_through_a_group_of_equal_scores(nf, levels)
    test_selection_with_more_cells_than_the_register_path_holds()
    test_selection_beyond_16k_cells(1080, 1920, 10000, 0.2, 1)
    test_selection_beyond_16k_cells(1088, 1920, 1000, 0.0, 3)
    test_postprocess_batch()
    from sp_orb_slam_amd import synth
    H, W, nf, B = 240, 376, 300, 3
    imgs = [synth.make_image(40 + i, H, W) for i in range(B)]
    ext = SPExtractor(nf, H, W, _blob(), max_batch=B)
    huge = ext.extract_batch(imgs)
    ext.close()
    monkeypatch.delenv("SPFE_SELECT_HUGE")
    ext = SPExtractor(nf, H, W, _blob(), max_batch=B)
    dflt = ext.extract_batch(imgs)
    assert ext.debug_read("select_huge")[0] == 0
    ext.close()
    for a, b in zip(huge, dflt):
        assert a.K == b.K and np.array_equal(a.kp_xy, b.kp_xy) and np.array_equal(a.occ_grid, b.occ_grid)
        assert np.array_equal(_bits(a.descriptors), _bits(b.descriptors)) and np.array_equal(_bits(a.cov2), _bits(b.cov2))


@pytest.mark.parametrize("env", [{"SPFE_COV_CAPS": ",,,,0"}, {"SPFE_COV_CAPS": ",,,,40"}, {}])
def test_covariance_link_from_the_classifications_edge_list_equals_the_pop_list_walk(monkeypatch, env):
    """Round 4: the classification lists the claim edges (lower claimant, dirty keypoint) while it has the claims in registers,
    and the link kernel unites from that list (one global round trip) and builds the chains by an all-pairs scan instead of a
    bitonic sort.  Against the pop-list walk (SPFE_COV_CAPS field 5, ecap = 0), and with a list too short for the frame (ecap = 40:
    the link kernel must notice and walk the pop lists): covariances bitwise equal to the sequential oracle's either way."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    H, W, nf = 240, 376, 400
    blob = weights.synthetic(7, "dense")
    imgs = [synth.make_image(210 + i, H, W) for i in range(2)]
    ext = SPExtractor(nf, H, W, blob, max_batch=2, with_heat=False)
    frs = ext.extract_batch(imgs)
    for fr, img in zip(frs, imgs):
        ref = oracle.extract(blob, img, nf)
        assert fr.K == ref["K"] and fr.status == 0
        assert np.array_equal(fr.cov2.view(np.uint32), ref["cov2"].view(np.uint32))
        assert np.array_equal(fr.cov2_inv.view(np.uint32), ref["cov2_inv"].view(np.uint32))
    nd = np.frombuffer(ext.debug_read("cov_counters", 0).tobytes(), np.int32)
    assert nd[0] > 10          # the frame has dirty keypoints: the link stage did work
    ext.close()


@pytest.mark.gpu
def test_replay_workers_are_listed_longest_chain_first():
    """cov_link_kernel lists the replay workers (first dirty member of each component) longest chain first, ties by lower
    keypoint: a replay workgroup takes consecutive workers and lives as long as its longest chain (cov.hip).  The order is
    deterministic and changes no result (the covariances are compared with the sequential oracle)."""
    H, W, nf = 480, 752, 1000
    blob = weights.synthetic(7, "dense")
    img = synth.make_image(200, H, W)
    ext = SPExtractor(nf, H, W, blob, with_heat=False)
    runs = []
    for _ in range(2):
        ext(img, None)
        cnt, nxt, workers = (ext.debug_read(n).copy() for n in ("cov_counters", "cov_nxt", "cov_workers"))
        runs.append(workers[:cnt[1]].copy())
        lens = []
        for w in workers[:cnt[1]]:
            j, n = int(w), 0
            while j >= 0:
                n += 1
                j = int(nxt[j])
            lens.append(n)
        assert cnt[1] > 20 and max(lens) >= 4, (cnt, lens[:8])
        assert all(a > b or (a == b and wa < wb) for a, b, wa, wb in zip(lens, lens[1:], workers, workers[1:])), lens
        assert sum(lens) == cnt[0]
    assert np.array_equal(runs[0], runs[1])
    ref = oracle.extract(blob, img, nf)
    assert np.array_equal(ext.last.cov2.view(np.uint32), ref["cov2"].view(np.uint32))
    ext.close()


def test_covariance_maps_keep_their_entries_across_batches(monkeypatch):
    """The claim / done maps are not cleared per batch: an entry carries its batch's generation code (counting down from
    32,766) and an older batch's reads as "nobody" (cov.hip).  Forced here to start at code 3 (SPFE_COV_CAPS field 6): the codes
    run out every third call, frames come and go between calls (a frame that sat out a wrap must be reset before its next use:
    its old entries' codes come round again), and a frame is redone by the last resort in between (it uses the claim map as
    its visited mask and must hand it back clean).  Every call's records equal a default handle's."""
    H, W, nf, B = 240, 376, 300, 3
    blob = _blob()
    from sp_orb_slam_amd import synth
    imgs = [synth.make_image(520 + i, H, W) for i in range(7)]
    plan = [[0, 1, 2], [3], [4], [5], [6, 0, 1], [2, 3], [4, 5, 6], [0], [1, 2, 3], [4, 5, 6]]
    ref_ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False)
    ref = [ref_ext.extract_batch([imgs[i] for i in p]) for p in plan]
    ref_ext.close()
    for caps in (",,,,,3", "24,0,,,,2"):      # (second: no overflow slots -> flagged frames go through cov_fallback_kernel; codes 2, 1, wrap)
        monkeypatch.setenv("SPFE_COV_CAPS", caps)
        ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False)
        for k, p in enumerate(plan):
            got = ext.extract_batch([imgs[i] for i in p])
            for g, e in zip(got, ref[k]):
                assert g.status == 0 and g.K == e.K and np.array_equal(g.kp_xy, e.kp_xy), (caps, k)
                assert np.array_equal(_bits(g.cov2), _bits(e.cov2)) and np.array_equal(_bits(g.cov2_inv), _bits(e.cov2_inv)), (caps, k)
                assert np.array_equal(_bits(g.response), _bits(e.response)), (caps, k)
        ext.close()
