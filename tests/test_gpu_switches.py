"""GPU: one bit-identity test per environment switch of libspfe.so that no other test file exercises (README.md
"Environment switches"): the handle is created with the NON-default value, driven the way the switch matters (synchronous
single-frame calls, or twelve pipelined steps on changing frames into separate record buffers), and every record must equal
the one a default synchronous handle computes for the same frames — every field, bit for bit.

The switches are read once per handle by spfe_create (read_switches, csrc/spfe_pack.hip), so monkeypatch.setenv before the
constructor is all it takes; the reference handle is made with the variable unset."""
import numpy as np
import pytest

from sp_orb_slam_amd import parallel, synth, weights
from sp_orb_slam_amd.extractor import SPExtractor

pytestmark = pytest.mark.gpu

FIELDS = ("kp_xy", "response", "descriptors", "cov2", "cov2_inv", "occ_grid", "dense_dust", "semi_dust")
ALL = ("SPFE_INLINE_CHAIN", "SPFE_DEFER_JOIN", "SPFE_TAIL_PER_HALF", "SPFE_TWO_CHAINS", "SPFE_SEL_EXT_EVENT", "SPFE_REPLAY_WAVES",
       "SPFE_ZERO_IN_TAIL", "SPFE_SPARSE_DA", "SPFE_PIPE_COPY_KERNEL", "SPFE_COMM_OWN_STREAM")


def _reference(torch, prec, H, W, B, nf, blob, sets):
    ext = SPExtractor(nf, H, W, blob, max_batch=B, precision=prec, with_heat=False)
    rb = ext.record_bytes()
    ref = []
    for d in sets:
        rec = torch.zeros(B * rb, dtype=torch.uint8, device="cuda")
        ext.extract_batch_device(d.data_ptr(), B, rec.data_ptr())
        torch.cuda.synchronize()
        ref.append([ext.view_record(rec.cpu().numpy()[i * rb:(i + 1) * rb]) for i in range(B)])
    ext.close()
    return ref, rb


def _same(a, b, where):
    assert a.status == 0 and a.K == b.K and a.K > 0, where
    for name in FIELDS:
        assert np.array_equal(getattr(a, name), getattr(b, name)), (where, name)


# (switch, value, precision, pipelined): pipelined switches act on steps that run as two half batches (f32, and bf16 frames
# of fewer than 10,000 cells) or on the side chain; synchronous ones on the single call's order
CASES = [
    ("SPFE_INLINE_CHAIN", "0", "f32", False), ("SPFE_INLINE_CHAIN", "0", "bf16", False),
    ("SPFE_SEL_EXT_EVENT", "0", "f32", False),
    ("SPFE_DEFER_JOIN", "0", "f32", True), ("SPFE_DEFER_JOIN", "0", "bf16", True),
    ("SPFE_TAIL_PER_HALF", "0", "f32", True), ("SPFE_TAIL_PER_HALF", "0", "bf16", True),
    ("SPFE_TWO_CHAINS", "1", "f32", True), ("SPFE_TWO_CHAINS", "1", "bf16", True),
    ("SPFE_REPLAY_WAVES", "8", "f32", True), ("SPFE_REPLAY_WAVES", "2", "bf16", True), ("SPFE_REPLAY_WAVES", "8", "f32", False),
    ("SPFE_ZERO_IN_TAIL", "0", "bf16", True), ("SPFE_ZERO_IN_TAIL", "0", "bf16", False),
    ("SPFE_SPARSE_DA", "2", "bf16", True), ("SPFE_SPARSE_DA", "0", "f32", True), ("SPFE_SPARSE_DA", "1", "f32", True),
]


@pytest.mark.parametrize("var,val,prec,pipelined", CASES)
def test_switch_gives_the_same_records(monkeypatch, var, val, prec, pipelined):
    import torch
    for v in ALL:
        monkeypatch.delenv(v, raising=False)
    H, W, B, nf, steps = 240, 376, 5, 300, 12
    blob = weights.synthetic(7, "dense")
    sets = [torch.from_numpy(np.stack([synth.make_image(640 + 7 * r + i, H, W) for i in range(B)])).cuda() for r in range(3)]
    ref, rb = _reference(torch, prec, H, W, B, nf, blob, sets)
    monkeypatch.setenv(var, val)
    ext = SPExtractor(nf, H, W, blob, max_batch=B, precision=prec, with_heat=False, async_cov=pipelined)
    recs = [torch.zeros(B * rb, dtype=torch.uint8, device="cuda") for _ in range(steps)]
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    for k in range(steps):
        ext.extract_batch_device(sets[k % 3].data_ptr(), B, recs[k].data_ptr(), stream.cuda_stream)
    if pipelined:
        ext.wait_records(ext.last_ticket(), stream.cuda_stream)
    torch.cuda.synchronize()
    for k in range(steps):
        host = recs[k].cpu().numpy()
        for i in range(B):
            _same(ext.view_record(host[i * rb:(i + 1) * rb]), ref[k % 3][i], (var, val, k, i))
    # single frames through the same handle (the inline chain / the selection's event act there)
    one = torch.zeros(rb, dtype=torch.uint8, device="cuda")
    for r in range(3):
        ext.extract_batch_device(sets[r].data_ptr(), 1, one.data_ptr(), stream.cuda_stream)
        if pipelined:
            ext.wait_records(ext.last_ticket(), stream.cuda_stream)
        torch.cuda.synchronize()
        _same(ext.view_record(one.cpu().numpy()), ref[r][0], (var, val, "single", r))
    ext.close()


@pytest.mark.parametrize("val,prec", [("0", "f32"), ("1", "bf16")])
def test_pipelined_host_path_copy_switch(monkeypatch, val, prec):
    """SPFE_PIPE_COPY_KERNEL: the D2H of a pipelined batch by the runtime's copy engine (0; the f32 default is the kernel) / by
    the library's copy kernel (1; the bf16 default for large frames is the engine): the same records reach the host."""
    monkeypatch.delenv("SPFE_PIPE_COPY_KERNEL", raising=False)
    H, W, B, nf = 120, 160, 2, 150
    blob = weights.synthetic(7, "dense")
    batches = [[synth.make_image(1200 + 10 * s + i, H, W) for i in range(B)] for s in range(4)]
    ref_ext = SPExtractor(nf, H, W, blob, max_batch=B, precision=prec, with_heat=False)
    ref = [ref_ext.extract_batch(b) for b in batches]
    ref_ext.close()
    monkeypatch.setenv("SPFE_PIPE_COPY_KERNEL", val)
    ext = SPExtractor(nf, H, W, blob, max_batch=B, precision=prec, with_heat=False)
    tk = [ext.submit_batch(b) for b in batches[:3]]
    got = []
    for k in range(4):
        got.append(ext.collect_batch(tk.pop(0)))
        if k == 0:
            tk.append(ext.submit_batch(batches[3]))
    ext.close()
    for k, (g, e) in enumerate(zip(got, ref)):
        for i in range(B):
            _same(g[i], e[i], ("copy", val, k, i))


def test_gather_on_a_communication_stream_of_its_own(monkeypatch):
    """SPFE_COMM_OWN_STREAM=1 (read by spfe_comm_init): the all-gather on a stream of its own that waits for the batch's
    covariance event instead of riding on the side stream — 1-rank RCCL communicator, pipelined driver, records equal the
    host call's.  The same handle then re-makes its communicator without the switch (bench.py's comm_stream_ab does that)."""
    import torch
    H, W, nf, B = 120, 160, 150, 3
    blob = weights.synthetic(7, "dense")
    batches = [np.stack([synth.make_image(1500 + 10 * s + i, H, W) for i in range(B)]) for s in range(3)]
    host = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False)
    expect = [host.extract_batch(list(b)) for b in batches]
    host.close()
    d_batches = [torch.from_numpy(b).cuda() for b in batches]
    ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, async_cov=True)
    streams = []
    for own in ("1", "0"):
        monkeypatch.setenv("SPFE_COMM_OWN_STREAM", own)
        sh = parallel.ShardedExtractor(ext, 1, 0, B, native_comm=True)
        streams.append(ext.comm_stream())
        comp = torch.cuda.Stream()
        done = []
        for k in range(6):
            sh.step(d_batches[k % 3], comp)
            if k >= 1:
                done.append((k - 1, [sh.decode(i) for i in range(B)]))
        sh.flush(comp)
        done.append((5, [sh.decode(i) for i in range(B)]))
        for k, got in done:
            for i, (g, e) in enumerate(zip(got, expect[k % 3])):
                _same(g, e, ("own_stream", own, k, i))
        ext.comm_destroy()
    assert streams[0] != streams[1]   # the own-stream communicator did not use the side stream
    ext.close()


def test_two_side_chains_in_flight_on_large_bf16_frames(monkeypatch):
    """bf16 frames of >= 10,000 cells, SPFE_FLAG_ASYNC_COV: the handle owns a twin (a second set of buffers and a second side
    stream) and pipelined device calls alternate between the two, so that the side chains of consecutive batches overlap
    (spfe_host.h).  Tickets stay one sequence; spfe_wait_records(t) still means "t and every earlier call" — checked WITHOUT a
    device-wide synchronisation: the stream that waited copies the record buffers out; records equal a synchronous handle's;
    debug reads come from whichever of the pair ran the last call; SPFE_TWO_CHAINS=0 gives the single chain."""
    import torch
    for v in ALL:
        monkeypatch.delenv(v, raising=False)
    H, W, B, nf, steps = 720, 1280, 3, 500, 7
    blob = weights.synthetic(7, "dense")
    sets = [torch.from_numpy(np.stack([synth.make_image(900 + 5 * r + i, H, W) for i in range(B)])).cuda() for r in range(3)]
    ref, rb = _reference(torch, "bf16", H, W, B, nf, blob, sets)
    for env, want in ((None, 1), ("0", 0)):
        if env is not None:
            monkeypatch.setenv("SPFE_TWO_CHAINS", env)
        ext = SPExtractor(nf, H, W, blob, max_batch=B, precision="bf16", with_heat=False, async_cov=True)
        assert int(ext.debug_read("two_chains")[0]) == want
        recs = [torch.zeros(B * rb, dtype=torch.uint8, device="cuda") for _ in range(steps)]
        stream = torch.cuda.Stream()
        torch.cuda.synchronize()
        tickets = []
        for k in range(steps):
            tickets.append(ext.extract_batch_device(sets[k % 3].data_ptr(), B, recs[k].data_ptr(), stream.cuda_stream))
        assert tickets == list(range(steps))
        ext.wait_records(tickets[-1], stream.cuda_stream)
        with torch.cuda.stream(stream):
            host = [r.to("cpu", non_blocking=False) for r in recs]    # (ordered on `stream` only)
        for k in range(steps):
            hk = host[k].numpy()
            for i in range(B):
                _same(ext.view_record(hk[i * rb:(i + 1) * rb]), ref[k % 3][i], ("two chains", env, k, i))
        # the last call's intermediates: semi of frame 0 equals a synchronous handle's
        semi = ext.debug_read("semi", 0)
        sref = SPExtractor(nf, H, W, blob, max_batch=B, precision="bf16", with_heat=False)
        one = torch.zeros(B * rb, dtype=torch.uint8, device="cuda")
        sref.extract_batch_device(sets[(steps - 1) % 3].data_ptr(), B, one.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(semi.view(np.uint32), sref.debug_read("semi", 0).view(np.uint32))
        # a synchronous host call in between runs on the handle itself: its intermediates are what a debug read returns then,
        # whichever of the pair ran the last pipelined call
        himgs = [synth.make_image(990 + i, H, W) for i in range(B)]
        got = ext.extract_batch(himgs)
        exp = sref.extract_batch(himgs)
        for i in range(B):
            _same(got[i], exp[i], ("two chains, host call", env, i))
        assert np.array_equal(ext.debug_read("semi", 1).view(np.uint32), sref.debug_read("semi", 1).view(np.uint32))
        sref.close()
        with pytest.raises(Exception):
            ext.wait_records(tickets[0], stream.cuda_stream)      # out of the window of the last four calls
        ext.close()


def test_two_side_chains_on_the_pipelined_host_path():
    """spfe_submit_batch / spfe_collect_batch on large bf16 frames: the twin is made at the first submission (the handle was
    created without SPFE_FLAG_ASYNC_COV), submissions alternate between the pair, each D2H copy rides the side stream of the
    handle that ran the batch, heat maps included; records and maps equal the synchronous calls'."""
    H, W, B, nf = 720, 1280, 2, 400
    blob = weights.synthetic(7, "dense")
    batches = [[synth.make_image(1500 + 10 * s + i, H, W) for i in range(B)] for s in range(6)]
    ref_ext = SPExtractor(nf, H, W, blob, max_batch=B, precision="bf16", with_heat=True)
    ref = [ref_ext.extract_batch(b) for b in batches]
    ref_ext.close()
    ext = SPExtractor(nf, H, W, blob, max_batch=B, precision="bf16", with_heat=True)
    assert int(ext.debug_read("two_chains")[0]) == 0
    tk = [ext.submit_batch(b) for b in batches[:3]]
    assert int(ext.debug_read("two_chains")[0]) == 1 and tk == [0, 1, 2]
    got = []
    for k in range(6):
        got.append(ext.collect_batch(tk.pop(0)))
        if k + 3 < 6:
            tk.append(ext.submit_batch(batches[k + 3]))
    ext.close()
    for k, (g, e) in enumerate(zip(got, ref)):
        for i in range(B):
            _same(g[i], e[i], ("pipe", k, i))
            assert np.array_equal(g[i].heat, e[i].heat) and np.array_equal(g[i].heat_inv, e[i].heat_inv), ("pipe heat", k, i)


def test_the_lazily_built_twin_does_not_read_the_callers_weight_memory(tmp_path, monkeypatch):
    """a handle created without SPFE_FLAG_ASYNC_COV builds its twin at the FIRST spfe_submit_batch — long after
    spfe_create returned and the caller's weight array / path string were released (SPExtractor drops both before __init__
    returns).  The library keeps its own parsed blob for the twin and the stored configuration points nowhere; the twin also
    runs its sibling's switches, not the environment of that later moment.  Weights from a path that is deleted and from a
    temporary array that is overwritten and freed; odd tickets (the twin's) must equal the synchronous records."""
    import gc
    for v in ALL:
        monkeypatch.delenv(v, raising=False)
    H, W, B, nf = 720, 1280, 2, 300
    blob = weights.synthetic(7, "dense")
    batches = [[synth.make_image(1700 + 10 * s + i, H, W) for i in range(B)] for s in range(4)]
    ref_ext = SPExtractor(nf, H, W, blob, max_batch=B, precision="bf16", with_heat=False)
    ref = [ref_ext.extract_batch(b) for b in batches]
    ref_ext.close()
    path = tmp_path / "w.spfw"
    weights.save(str(path), blob)
    for how in ("path", "array"):
        if how == "path":
            ext = SPExtractor(nf, H, W, str(path), max_batch=B, precision="bf16", with_heat=False)
            path.unlink()                                   # the file is gone before the twin is built
        else:
            tmp = blob.copy()
            ext = SPExtractor(nf, H, W, tmp, max_batch=B, precision="bf16", with_heat=False)
            tmp[:] = np.float32(np.nan)                     # the caller's memory is garbage, then released
            del tmp
            gc.collect()
            junk = [np.full(blob.size, 7.0, np.float32) for _ in range(4)]   # (likely to land where the array was)
            assert junk
        monkeypatch.setenv("SPFE_SPARSE_DA", "0")           # an environment change behind spfe_create: the twin must not see it
        monkeypatch.setenv("SPFE_TWO_CHAINS", "0")
        assert int(ext.debug_read("two_chains")[0]) == 0
        tk = [ext.submit_batch(b) for b in batches[:3]]
        assert int(ext.debug_read("two_chains")[0]) == 1
        got = [ext.collect_batch(t) for t in tk]
        got.append(ext.collect_batch(ext.submit_batch(batches[3])))
        assert int(ext.debug_read("da_gathered")[0]) == 1   # (ticket 3 ran on the twin: gathered convDa = the sibling's switch)
        ext.close()
        monkeypatch.delenv("SPFE_SPARSE_DA")
        monkeypatch.delenv("SPFE_TWO_CHAINS")
        for k, (g, e) in enumerate(zip(got, ref)):
            for i in range(B):
                _same(g[i], e[i], ("twin lifetime", how, k, i))


def test_a_twin_that_cannot_be_built_is_not_fatal(monkeypatch):
    """the twin is an optimisation — when its build fails (a second full set of buffers), spfe_create / the first
    spfe_submit_batch carry on with one side chain.  SPFE_TWO_CHAINS=99 wants a twin and makes its build fail."""
    import torch
    for v in ALL:
        monkeypatch.delenv(v, raising=False)
    H, W, B, nf = 240, 376, 2, 200
    blob = weights.synthetic(7, "dense")
    sets = [torch.from_numpy(np.stack([synth.make_image(1800 + 5 * r + i, H, W) for i in range(B)])).cuda() for r in range(3)]
    ref, rb = _reference(torch, "f32", H, W, B, nf, blob, sets)
    monkeypatch.setenv("SPFE_TWO_CHAINS", "99")
    ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, async_cov=True)      # spfe_create: twin wanted, fails, handle lives
    assert int(ext.debug_read("two_chains")[0]) == 0 and int(ext.debug_read("twin_failed")[0]) == 1
    recs = [torch.zeros(B * rb, dtype=torch.uint8, device="cuda") for _ in range(5)]
    stream = torch.cuda.Stream()
    for k in range(5):
        t = ext.extract_batch_device(sets[k % 3].data_ptr(), B, recs[k].data_ptr(), stream.cuda_stream)
    ext.wait_records(t, stream.cuda_stream)
    stream.synchronize()
    for k in range(5):
        hk = recs[k].cpu().numpy()
        for i in range(B):
            _same(ext.view_record(hk[i * rb:(i + 1) * rb]), ref[k % 3][i], ("twin failed", k, i))
    ext.close()
    ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False)                      # the lazy form: first spfe_submit_batch
    himgs = [[synth.make_image(1800 + 5 * r + i, H, W) for i in range(B)] for r in range(3)]
    got = [ext.collect_batch(ext.submit_batch(b)) for b in himgs]
    assert int(ext.debug_read("two_chains")[0]) == 0 and int(ext.debug_read("twin_failed")[0]) == 1
    ext.close()
    for k in range(3):
        for i in range(B):
            _same(got[k][i], ref[k][i], ("twin failed, host path", k, i))


def test_two_chains_into_one_record_buffer_stay_ordered(monkeypatch):
    """with a twin, consecutive pipelined calls run on different handles; a caller that passes the SAME record
    buffer call after call is still ordered — call i + 1's tail and chain do not write the buffer while chain i, on the
    sibling's side stream, is still writing it.  30 calls into one buffer, frames alternating: the buffer holds the last
    call's records, bit for bit."""
    import torch
    for v in ALL:
        monkeypatch.delenv(v, raising=False)
    H, W, B, nf = 240, 376, 3, 300
    blob = weights.synthetic(7, "dense")
    sets = [torch.from_numpy(np.stack([synth.make_image(1900 + 5 * r + i, H, W) for i in range(B)])).cuda() for r in range(2)]
    for prec in ("f32", "bf16"):
        ref, rb = _reference(torch, prec, H, W, B, nf, blob, sets)
        monkeypatch.setenv("SPFE_TWO_CHAINS", "1")
        ext = SPExtractor(nf, H, W, blob, max_batch=B, precision=prec, with_heat=False, async_cov=True)
        monkeypatch.delenv("SPFE_TWO_CHAINS")
        assert int(ext.debug_read("two_chains")[0]) == 1
        rec = torch.zeros(B * rb, dtype=torch.uint8, device="cuda")
        stream = torch.cuda.Stream()
        for last in (29, 30):
            for k in range(last + 1):
                t = ext.extract_batch_device(sets[k & 1].data_ptr(), B, rec.data_ptr(), stream.cuda_stream)
            ext.wait_records(t, stream.cuda_stream)
            stream.synchronize()
            hk = rec.cpu().numpy()
            for i in range(B):
                _same(ext.view_record(hk[i * rb:(i + 1) * rb]), ref[last & 1][i], ("one buffer", prec, last, i))
        ext.close()
