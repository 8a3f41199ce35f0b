"""GPU: the gathered descriptor head ("sparse convDb").  The descriptor head (convDb, sp_extractor.cpp:99-100) feeds
only the bilinear sampling at the emitted keypoints (:134-148), so the product path runs it BEHIND the selection, on
the coarse cells some keypoint's four taps read, through a cell list the selection kernel writes.  Checked here:
  * the list is exactly the union of the taps (recomputed in numpy from the record's keypoints),
  * the rows it wrote hold the dense head's bits,
  * records are the same bits with SPFE_SPARSE_DB=0 and =1, synchronous and pipelined, f32 and bf16
(the f32 records' parity with the oracle is test_gpu_parity.py's, which runs the default = sparse path)."""
import numpy as np
import pytest

from sp_orb_slam_amd import synth, weights
from sp_orb_slam_amd.extractor import SPExtractor

pytestmark = pytest.mark.gpu
f32 = np.float32


def _bits(a):
    return np.ascontiguousarray(a, f32).view(np.uint32)


def _tap_cells(kp_xy, H, W):
    """Cells (row-major on the H/8 x W/8 coarse map) that grid_sample's bilinear taps read for these keypoints:
    sp_extractor.cpp:137-138 + align_corners un-normalisation, in float32 like the device."""
    wc, hc = W // 8, H // 8
    x, y = kp_xy[:, 0].astype(f32), kp_xy[:, 1].astype(f32)
    inv_hw, inv_hh = f32(1.0 / float(f32(W / 2.0))), f32(1.0 / float(f32(H / 2.0)))
    gx, gy = x * inv_hw - f32(1), y * inv_hh - f32(1)
    ix = ((gx + f32(1)) / f32(2)) * f32(wc - 1)
    iy = ((gy + f32(1)) / f32(2)) * f32(hc - 1)
    x0, y0 = np.floor(ix).astype(np.int64), np.floor(iy).astype(np.int64)
    cells = set()
    for dx in (0, 1):
        for dy in (0, 1):
            xx, yy = x0 + dx, y0 + dy
            ok = (xx >= 0) & (xx < wc) & (yy >= 0) & (yy < hc)
            cells.update((yy[ok] * wc + xx[ok]).tolist())
    return cells


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("H,W,B,nf", [(240, 376, 2, 300), (480, 752, 3, 1000), (720, 1280, 2, 1000), (120, 160, 1, 1000),
                                      (136, 200, 2, 50)])
def test_cell_list_is_the_union_of_the_taps_and_its_rows_are_the_dense_bits(monkeypatch, precision, H, W, B, nf):
    blob = weights.synthetic(7, "dense")
    imgs = [synth.make_image(70 + i, H, W) for i in range(B)]
    C = (H // 8) * (W // 8)
    monkeypatch.setenv("SPFE_SPARSE_DB", "0")       # the dense launches (convPa|Da, convDb) of another handle
    ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, precision=precision)
    ext.extract_batch(imgs)
    dense = [ext.debug_read("coarse", i).reshape(C, 256).copy() for i in range(B)]
    ext.close()
    monkeypatch.setenv("SPFE_SPARSE_DB", "1")
    ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, precision=precision)
    frs = ext.extract_batch(imgs)
    raw = [ext.debug_read("coarse_sparse", i).reshape(C, 256).copy() for i in range(B)]
    total = int(ext.debug_read("db_total")[0])
    cells = ext.debug_read("db_list")[:total].copy()
    full = [ext.debug_read("coarse", i).reshape(C, 256).copy() for i in range(B)]   # (completes the map on demand)
    ext.close()
    for b in range(B):
        assert np.array_equal(_bits(full[b]), _bits(dense[b]))
    assert len(set(cells.tolist())) == total
    for b in range(B):
        mine = np.sort(cells[(cells >= b * C) & (cells < (b + 1) * C)] - b * C)
        want = _tap_cells(frs[b].kp_xy, H, W)
        assert set(mine.tolist()) == want, (b, len(mine), len(want))
        assert np.array_equal(_bits(raw[b][mine]), _bits(dense[b][mine]))
        assert len(want) <= 4 * frs[b].K


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("H,W,B", [(240, 376, 3), (720, 1280, 2), (128, 168, 4), (480, 752, 3)])
def test_records_do_not_depend_on_the_descriptor_head_being_gathered(monkeypatch, precision, H, W, B):
    """Synchronous calls (twice: the second runs behind the first's side chain) and the pipelined host path, then a
    synchronous single frame again — on ONE handle.  SPFE_SPARSE_DB = 0 never / 1 every call / 2 synchronous calls only (the
    default of bf16 frames below 10,000 cells since round 4: gathered and dense calls alternate on the handle) / unset."""
    blob = weights.synthetic(7, "dense")
    sets = [[synth.make_image(90 + 10 * r + i, H, W) for i in range(B)] for r in range(3)]
    out = {}
    for flag in ("0", "1", "2", None):
        if flag is None:
            monkeypatch.delenv("SPFE_SPARSE_DB", raising=False)
            flag = "default"
        else:
            monkeypatch.setenv("SPFE_SPARSE_DB", flag)
        ext = SPExtractor(500, H, W, blob, max_batch=B, with_heat=False, precision=precision)
        sync = [ext.extract_batch(s) for s in sets[:2]]
        tickets = [ext.submit_batch(s) for s in sets]
        pipe = [ext.collect_batch(t) for t in tickets]
        one = ext.extract_batch(sets[2][:1])[0]
        ext.close()
        out[flag] = [fr for frs in sync + pipe for fr in frs] + [one]
    assert len(out["0"]) == 5 * B + 1
    for a, b in [ab for other in ("1", "2", "default") for ab in zip(out["0"], out[other])]:
        assert a.K == b.K and a.status == 0 and b.status == 0
        assert np.array_equal(a.kp_xy, b.kp_xy)
        assert np.array_equal(_bits(a.descriptors), _bits(b.descriptors))
        assert np.array_equal(_bits(a.response), _bits(b.response))
        assert np.array_equal(_bits(a.cov2), _bits(b.cov2))


def test_no_keypoints_no_cells(monkeypatch):
    """A black frame has no candidates: the list is empty, the gathered launch walks nothing, the record is empty."""
    monkeypatch.setenv("SPFE_SPARSE_DB", "1")
    H, W = 120, 160
    blob = weights.synthetic(7, "sparse")
    ext = SPExtractor(100, H, W, blob, with_heat=False)
    ext.extract_batch([synth.make_image(5, H, W)])          # something first, so that the counters are not just their initial zeros
    fr = ext.extract_batch([np.zeros((H, W), np.uint8)])[0]
    total = int(ext.debug_read("db_total")[0])
    ext.close()
    assert total == len(_tap_cells(fr.kp_xy, H, W))
    if fr.K == 0:
        assert total == 0


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_gathered_branch_with_changing_batch_sizes_and_a_full_list(monkeypatch, precision):
    """Calls of 3, 1, 2, 3 frames on one handle (the list, its total and the parity-double-buffered conv4b output are per
    call), and num_features so large that the list holds every cell (4 kmax > C: capacity C)."""
    H, W = 128, 168
    blob = weights.synthetic(7, "dense")
    sets = [[synth.make_image(40 + 7 * r + i, H, W) for i in range(n)] for r, n in enumerate((3, 1, 2, 3))]
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("SPFE_SPARSE_DB", flag)
        res = []
        for nf in (120, 2000):
            ext = SPExtractor(nf, H, W, blob, max_batch=3, with_heat=False, precision=precision)
            for s in sets:
                res += ext.extract_batch(s)
            tickets = [ext.submit_batch(s) for s in sets[:3]]
            for t in tickets:
                res += ext.collect_batch(t)
            ext.close()
        out[flag] = res
    assert len(out["0"]) == len(out["1"]) == 2 * (9 + 6)
    for a, b in zip(out["0"], out["1"]):
        assert a.K == b.K and np.array_equal(a.kp_xy, b.kp_xy)
        assert np.array_equal(_bits(a.descriptors), _bits(b.descriptors))


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_gathered_branch_at_1920x1080(monkeypatch, precision):
    """32,400 cells: the selection's large-frame path writes the list, the gathered kernels' 32-bit offsets hold."""
    H, W = 1080, 1920
    blob = weights.synthetic(7, "dense")
    img = synth.make_image(11, H, W)
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("SPFE_SPARSE_DB", flag)
        ext = SPExtractor(1000, H, W, blob, with_heat=False, precision=precision)
        out[flag] = ext.extract_batch([img])[0]
        if flag == "1":
            total = int(ext.debug_read("db_total")[0])
            assert total == len(_tap_cells(out[flag].kp_xy, H, W))
        ext.close()
    a, b = out["0"], out["1"]
    assert a.K == b.K and np.array_equal(a.kp_xy, b.kp_xy)
    assert np.array_equal(_bits(a.descriptors), _bits(b.descriptors))
    assert np.array_equal(_bits(a.cov2), _bits(b.cov2))
