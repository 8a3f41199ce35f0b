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


def _worker(rank, world, port, q):
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


def test_two_rank_gloo_gather_matches_single_process():
    from oracle import oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(WORLD))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    blob = weights.synthetic(7, "dense")
    expect = []
    for fidx in range(NFRAMES):
        o = oracle.extract(blob, synth.make_image(40 + fidx, H, W), NF)
        expect.append((fidx, o["K"], o["kp_xy"].tobytes(), o["desc"].tobytes(), o["occ_grid"].tobytes(),
                       o["cov2_inv"].tobytes()))
    for r in range(WORLD):
        assert results[r] == expect      # every rank holds every frame, in global order
