"""CPU, world_size 2, gloo: the N > 1 path — shard a global batch, produce the
per-frame records on each rank (here from the oracle, packed with the Python
mirror of the record layout), all-gather them, decode every frame on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sp_orb_slam_amd import parallel, synth, weights

H, W, NF, NFRAMES, WORLD = 64, 96, 30, 5, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, NFRAMES=NFRAMES):
    from oracle import oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lay = parallel.RecordLayout(H, W, NF)
        blob = weights.synthetic(7, "dense")
        lo, hi = parallel.shard_range(NFRAMES, world, rank)
        per_rank = -(-NFRAMES // world)  # pad shards to equal length
        local = np.zeros((per_rank, lay.bytes), np.uint8)
        for i, fidx in enumerate(range(lo, hi)):
            out = oracle.extract(blob, synth.make_image(40 + fidx, H, W), NF)
            out["status"] = 0
            local[i] = lay.pack(out)
        gathered = parallel.gather_records(torch.from_numpy(local.reshape(-1)), world).numpy()
        gathered = gathered.reshape(world, per_rank, lay.bytes)
        got = []
        for r in range(world):
            rlo, rhi = parallel.shard_range(NFRAMES, world, r)
            for i in range(rhi - rlo):
                d = lay.unpack(gathered[r, i])
                got.append((rlo + i, d["K"], d["kp_xy"].tobytes(), d["desc"].tobytes(), d["occ_grid"].tobytes(),
                            d["cov2_inv"].tobytes()))
        q.put((rank, got))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions():
    for n in (0, 1, 5, 8, 64, 67):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_range(4, 2, 2)


def test_record_codec_roundtrip():
    from oracle import oracle
    lay = parallel.RecordLayout(H, W, NF)
    assert lay.bytes % 256 == 0 and lay.off_desc % 16 == 0
    out = oracle.extract(weights.synthetic(7, "dense"), synth.make_image(3, H, W), NF)
    out["status"] = 0
    d = lay.unpack(lay.pack(out))
    assert d["K"] == out["K"] and np.array_equal(d["kp_xy"], out["kp_xy"])
    assert np.array_equal(d["desc"], out["desc"]) and np.array_equal(d["occ_grid"], out["occ_grid"])
    assert np.array_equal(d["cov2_inv"], out["cov2_inv"]) and np.array_equal(d["dense_dust"], out["dense_dust"])
    bad = lay.pack(out)
    bad[lay.off_hdr:lay.off_hdr + 4] = np.array([lay.kmax + 5], np.int32).view(np.uint8)
    with pytest.raises(ValueError):
        lay.unpack(bad)


@pytest.mark.parametrize("NFRAMES,WORLD", [(5, 2), (64, 8), (67, 8)])
def test_gloo_gather_matches_single_process(NFRAMES, WORLD):
    """(5, 2): ragged shards on two ranks.  (64, 8): BASELINE configs[2] literally — 64 frames, 8 per rank, 8 ranks — as a dry
    run on small frames: `shard_range` for 8, the 64-record gathered buffer, ALL 64 records on every rank (rank 7 included)
    against the single-process result.  (67, 8): the same with ragged shards (9 / 8 frames)."""
    from oracle import oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q, NFRAMES)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(WORLD))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    blob = weights.synthetic(7, "dense")
    expect = []
    for fidx in range(NFRAMES):
        o = oracle.extract(blob, synth.make_image(40 + fidx, H, W), NF)
        expect.append((fidx, o["K"], o["kp_xy"].tobytes(), o["desc"].tobytes(), o["occ_grid"].tobytes(),
                       o["cov2_inv"].tobytes()))
    assert sorted(results) == list(range(WORLD))
    for r in range(WORLD):
        assert len(results[r]) == NFRAMES
        assert results[r] == expect      # every rank holds every frame, in global order


class _StubExtractor:
    """What init_native_comm touches of an SPExtractor: the id, the init (failing on the ranks listed), the teardown."""

    def __init__(self, rank, fail_ranks):
        self.rank, self.fail_ranks, self.inited, self.destroyed, self.uid_seen = rank, fail_ranks, False, 0, None

    def comm_unique_id(self):
        return bytes(range(128))

    def comm_init(self, uid, rank, world):
        self.uid_seen = uid
        if rank in self.fail_ranks:
            raise RuntimeError("ncclCommInitRank refused (injected)")
        self.inited = True

    def comm_destroy(self):
        self.destroyed += 1
        self.inited = False


def _comm_worker(rank, world, port, q, fail_ranks):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ext = _StubExtractor(rank, fail_ranks)
        ok, err = parallel.init_native_comm(ext, world, rank, device="cpu")
        q.put((rank, ok, err, ext.inited, ext.destroyed, ext.uid_seen == bytes(range(128))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fail_ranks", [(1,), (0,), ()])
def test_one_rank_failing_comm_init_sends_every_rank_to_the_same_fallback(fail_ranks):
    """spfe_comm_init fails on ONE rank (injected) -> the min-reduce makes EVERY rank report "no native
    communicator", the ranks whose init had succeeded destroy theirs, and all of them take torch's all-gather; with no
    failure every rank keeps its communicator.  The 128-byte id reaches rank 1 through the broadcast either way."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, WORLD, port, q, tuple(fail_ranks))) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in range(WORLD))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect_ok = not fail_ranks
    for r in range(WORLD):
        ok, err, inited, destroyed, uid_ok = res[r]
        assert ok == expect_ok and uid_ok
        assert (err is not None) == (r in fail_ranks)
        assert inited == expect_ok                       # nobody is left holding a communicator the others do not have
        assert destroyed == (0 if expect_ok else 1)
