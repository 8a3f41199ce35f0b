"""Frame-sharded multi-GPU extraction: one process per GPU, RCCL all-gather of
fixed-stride result records.

The reference is single-GPU, batch 1 (sp_extractor.cpp:70 "TODO: batch-size",
.squeeze() at :97); frames are independent (operator() reads only the image and
the weights), so a batch shards contiguously over ranks with NO collective in
the data path.  The only exchange is the final gather of what every consumer
needs: keypoints, descriptors, covariances, occ_grid and the dustbin maps, packed
by the kernels into one record per frame (layout: spfe_record_layout in
include/spfe.h).  Records have a fixed stride, so one `all_gather_into_tensor`
moves everything and no count exchange is needed (K lives in the header).

xGMI is point-to-point (7 links per GPU): with 8 frames x ~1.1 MB per rank the
gather is ~9 MB per rank, far below a millisecond per link — the collective is
not the bottleneck, so it is issued once per batch, not per frame.

`torch` is used for device buffers, streams and the process group only.  The
gather and the record codec also run on CPU tensors with the gloo backend,
which is how tests cover the N > 1 path without GPUs.
"""
import numpy as np

DESC_DIM = 256


def shard_range(n_frames, world, rank):
    """Contiguous split of a global batch: frames [lo, hi) belong to `rank`.
    The first n_frames % world ranks get one extra frame."""
    if world < 1 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad shard arguments")
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _align(v, a):
    return (v + a - 1) // a * a


class RecordLayout:
    """Python mirror of make_layout() in csrc/spfe_pack.hip (checked against the
    library by tests): byte offsets inside one per-frame record."""

    def __init__(self, height, width, num_features, desc_bf16=False):
        self.height, self.width = height, width
        self.desc_bf16 = bool(desc_bf16)   # SPFE_FLAG_DESC_BF16: 2-byte descriptor elements
        self.kmax = num_features + 1
        self.cells = (height // 8) * (width // 8)
        o = 0
        self.off_hdr = o; o += 16
        self.off_xy = o; o = _align(o + self.kmax * 2 * 4, 16)
        self.off_resp = o; o = _align(o + self.kmax * 4, 16)
        self.off_cov = o; o = _align(o + self.kmax * 2 * 4, 16)
        self.off_cinv = o; o = _align(o + self.kmax * 2 * 4, 16)
        self.off_desc = o; o = _align(o + self.kmax * DESC_DIM * (2 if self.desc_bf16 else 4), 16)
        self.off_occ = o; o = _align(o + self.cells * 2, 16)
        self.off_dd = o; o = _align(o + self.cells * 4, 16)
        self.off_sd = o; o = _align(o + self.cells * 4, 16)
        self.bytes = _align(o, 256)

    # -- codec (numpy, no GPU, no library) --
    def pack(self, fr):
        """Dict/obj with K, n_candidates, kp_xy, response, cov2, cov2_inv, desc,
        occ_grid, dense_dust, semi_dust -> uint8[bytes]."""
        g = (lambda k: fr[k]) if isinstance(fr, dict) else (lambda k: getattr(fr, k))
        K = int(g("K"))
        if K > self.kmax:
            raise ValueError("K=%d exceeds kmax=%d" % (K, self.kmax))
        rec = np.zeros(self.bytes, np.uint8)
        rec[self.off_hdr:self.off_hdr + 16] = np.array(
            [K, int(g("n_candidates")), int(g("status")) if self._has(fr, "status") else 0, 0],
            np.int32).view(np.uint8)

        def put(off, arr, dt):
            b = np.ascontiguousarray(arr, dt).reshape(-1).view(np.uint8)
            rec[off:off + b.size] = b

        put(self.off_xy, g("kp_xy"), np.float32)
        put(self.off_resp, g("response"), np.float32)
        put(self.off_cov, g("cov2"), np.float32)
        put(self.off_cinv, g("cov2_inv"), np.float32)
        d = g("descriptors") if self._has(fr, "descriptors") else g("desc")
        if self.desc_bf16:
            u = np.ascontiguousarray(d, np.float32).view(np.uint32).astype(np.uint64)
            put(self.off_desc, ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16), np.uint16)   # round to nearest even
        else:
            put(self.off_desc, d, np.float32)
        put(self.off_occ, g("occ_grid"), np.int16)
        put(self.off_dd, g("dense_dust"), np.float32)
        put(self.off_sd, g("semi_dust"), np.float32)
        return rec

    @staticmethod
    def _has(fr, k):
        return (k in fr) if isinstance(fr, dict) else hasattr(fr, k)

    def unpack(self, rec):
        """uint8[bytes] -> dict (views are copied)."""
        rec = np.ascontiguousarray(rec, np.uint8)
        if rec.size != self.bytes:
            raise ValueError("record has %d bytes, expected %d" % (rec.size, self.bytes))
        hdr = rec[self.off_hdr:self.off_hdr + 16].view(np.int32)
        K = int(hdr[0])
        if not (0 <= K <= self.kmax):
            raise ValueError("corrupt record: K=%d" % K)
        hc, wc = self.height // 8, self.width // 8

        def get(off, n, dt):
            return rec[off:off + n * np.dtype(dt).itemsize].view(dt).copy()

        return dict(K=K, n_candidates=int(hdr[1]), status=int(hdr[2]),
                    kp_xy=get(self.off_xy, K * 2, np.float32).reshape(K, 2),
                    response=get(self.off_resp, K, np.float32),
                    cov2=get(self.off_cov, K * 2, np.float32).reshape(K, 2),
                    cov2_inv=get(self.off_cinv, K * 2, np.float32).reshape(K, 2),
                    desc=((get(self.off_desc, K * DESC_DIM, np.uint16).astype(np.uint32) << 16).view(np.float32)
                          if self.desc_bf16 else get(self.off_desc, K * DESC_DIM, np.float32)).reshape(K, DESC_DIM),
                    occ_grid=get(self.off_occ, hc * wc, np.int16).reshape(hc, wc),
                    dense_dust=get(self.off_dd, hc * wc, np.float32).reshape(hc, wc),
                    semi_dust=get(self.off_sd, hc * wc, np.float32).reshape(hc, wc))


def gather_records(local_records, world, group=None):
    """all-gather the per-rank record blocks.

    local_records: torch uint8 tensor [frames_per_rank * record_bytes] (every rank
    the same length: pad the shard if the batch does not divide evenly).
    Returns [world * frames_per_rank * record_bytes], rank-major = global frame
    order of shard_range().  On the GPU this is one RCCL all-gather over xGMI
    (backend "nccl"); on CPU tensors it runs over gloo.
    """
    import torch
    import torch.distributed as dist

    if world == 1:
        return local_records
    out = torch.empty(world * local_records.numel(), dtype=torch.uint8, device=local_records.device)
    dist.all_gather_into_tensor(out, local_records.contiguous(), group=group)
    return out


def init_native_comm(extractor, world, rank, device="cuda"):
    """The library's RCCL communicator on `extractor` (spfe_comm_unique_id on rank 0, broadcast of the 128-byte id through
    torch.distributed, spfe_comm_init on every rank) — and the AGREEMENT that follows: a rank whose spfe_comm_init failed
    (librccl missing, ncclCommInitRank refused) and a rank where it succeeded must not part ways, one calling ncclAllGather
    and the other torch's all-gather.  So the outcome is min-reduced over the ranks; if any rank failed, EVERY rank destroys
    its communicator and the caller falls back to torch.distributed's all-gather on all of them.  Returns (ok, error text
    of this rank or None).  `device`: where the id / flag tensors live ("cuda" under RCCL; "cpu" in the gloo tests)."""
    import torch

    uid = torch.zeros(128, dtype=torch.uint8, device=device)
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(extractor.comm_unique_id()), dtype=torch.uint8))
    if world > 1:
        import torch.distributed as dist

        dist.broadcast(uid, 0)
    ok, err = 1, None
    try:
        extractor.comm_init(bytes(uid.cpu().numpy().tobytes()), rank, world)
    except Exception as e:   # RCCL missing / init failure on this rank
        ok, err = 0, str(e)
    if world > 1:
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag.item())
    if not ok:
        try:
            extractor.comm_destroy()
        except Exception:
            pass
    return bool(ok), err


class ShardedExtractor:
    """Per-rank driver of the batched path (BASELINE configs[2]): takes this rank's shard of
    device-resident frames, runs the HIP path, all-gathers the records.  One instance per
    process / GPU.

    Streams.  Compute runs on the caller's `stream`; with world > 1 the collective runs on a
    dedicated communication stream that waits only for the records of the batch it gathers
    (spfe_wait_records), never for younger compute — so the all-gather of batch i-1 overlaps the
    convolutions of batch i instead of queueing behind them, and nothing goes through the
    legacy default stream (whose implicit barriers would serialise compute and collective).
    Record buffers rotate (3 deep when pipelined) so a batch being gathered is never the one
    being written; gather outputs alternate between two buffers.

    With an extractor built with async_cov=True the driver is software pipelined: step(i)
    enqueues the compute of batch i and then completes batch i-1 (its covariance ran beside
    batch i's convolutions); flush() completes the last batch.  `gathered` is the most recently
    completed batch; call sync(stream) (or decode()) before consuming it.

    The collective itself is the LIBRARY's: spfe_allgather_records (ncclAllGather on a library-owned
    communication stream, include/spfe.h) — the same entry point the C++ SLAM host uses — set up by
    `native_comm=True` (the default when world > 1 and the process group is RCCL: the 128-byte unique id is
    created on rank 0 and broadcast through torch.distributed, the only thing torch still does for the
    gather).  native_comm=True with world == 1 gives a 1-rank communicator (self-gather; used by tests).
    gather_fn(out, local) replaces the collective (tests: a device copy, to exercise the stream / buffer
    logic on one GPU); with neither, torch.distributed.all_gather_into_tensor is the fallback (gloo dry
    runs on CPU-only / shared-GPU boxes)."""

    def __init__(self, extractor, world, rank, frames_per_rank, gather_fn=None, native_comm=None):
        import torch

        self.ext, self.world, self.rank, self.fpr = extractor, world, rank, frames_per_rank
        self.rec_bytes = extractor.record_bytes()
        self._gather_fn = gather_fn
        if native_comm is None:
            native_comm = False
            if world > 1 and gather_fn is None:
                import torch.distributed as dist

                native_comm = dist.is_initialized() and dist.get_backend() == "nccl"
        import os

        if os.environ.get("SPFE_NATIVE_COMM") == "0":   # force torch.distributed's all-gather
            native_comm = False
        self._native = bool(native_comm) and gather_fn is None
        self._collective = world > 1 or gather_fn is not None or self._native
        nbuf = 3 if extractor.async_cov else (2 if self._collective else 1)
        nbytes = frames_per_rank * self.rec_bytes
        self.local = [torch.zeros(nbytes, dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
        self._local_free = [None] * nbuf          # event: the gather that last read local[k] is done
        self.all = ([torch.zeros(world * nbytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
                    if self._collective else None)
        if self._native:
            self._native, self.native_comm_error = init_native_comm(extractor, world, rank, "cuda")
            if not self._native and world == 1:
                raise RuntimeError("spfe_comm_init failed: %s" % (self.native_comm_error or "?"))
        if self._native:
            # the library's communication stream, wrapped so torch events can be recorded on it
            self.comm = torch.cuda.ExternalStream(extractor.comm_stream())
        else:
            self.comm = torch.cuda.Stream() if self._collective else None
        self.gathered = None
        self._gathered_event = None
        self._pending = None
        self._lazy_ticket = None
        self._consumer = None
        self._calls = 0
        self._gathers = 0

    def _complete(self, ticket, k, stream):
        import os

        import torch

        if not self._collective:
            # no gather: `gathered` is this rank's own buffer, complete when the library's covariance event of that ticket has
            # fired.  Nothing is put on the compute stream here: a wait + an event record per step are two barrier packets
            # between two steps — ~20 us of an idle compute queue on a kernel timeline (round 4) — and nobody needs the
            # ordering until a consumer shows up: sync() / decode() / flush() establish it then.  (Buffer reuse is ordered by
            # the library itself: a call's detector tail waits for the side chain two tickets back.)
            self.gathered, self._gathered_event = self.local[k], None
            self._lazy_ticket = ticket
            if os.environ.get("SPFE_LAZY_ORDER") == "0":   # A/B knob: the per-step wait + record on the compute stream
                self._settle(stream)
            return
        out = self.all[self._gathers % 2]
        self._gathers += 1
        if self._native:
            # spfe_allgather_records: waits (on the communication stream) for exactly this batch's records,
            # covariance included, then ncclAllGather
            self.ext.allgather_records(ticket, self.local[k].data_ptr(), out.data_ptr(), self.fpr)
            ev = torch.cuda.Event()
            ev.record(self.comm)
            self._local_free[k] = ev
            self.gathered, self._gathered_event = out, ev
            return
        # the communication stream waits for exactly this batch's records (covariance included)
        self.ext.wait_records(ticket, self.comm.cuda_stream)
        with torch.cuda.stream(self.comm):
            if self._gather_fn is not None:
                self._gather_fn(out, self.local[k])
            else:
                import torch.distributed as dist

                dist.all_gather_into_tensor(out, self.local[k])
            ev = torch.cuda.Event()
            ev.record(self.comm)
        self._local_free[k] = ev
        self.gathered, self._gathered_event = out, ev

    def step(self, d_images, stream):
        """d_images: torch uint8 [frames_per_rank, H, W] on this rank's GPU; `stream`: torch.cuda.Stream
        the compute is enqueued on (a non-default stream when world > 1).

        What `self.gathered` holds afterwards is COMPLETE ONLY AFTER sync() / decode() / flush(): with one rank and
        pipelined calls nothing is put on `stream` per step (no wait, no event record: ~20 us of idle compute queue per
        step otherwise), so a consumer that reads `gathered` on `stream` right after step() races with the side chain —
        call sync(stream) first (it orders `stream` behind the batch), or set SPFE_LAZY_ORDER=0 for the per-step ordering."""
        k = self._calls % len(self.local)
        self._calls += 1
        if self._local_free[k] is not None:      # (long done: it was issued len(local) steps ago)
            stream.wait_event(self._local_free[k])
            self._local_free[k] = None
        buf = self.local[k]
        if self.ext.async_cov and self._native and self._pending is not None:
            # library gather: enqueued BEFORE this batch's work — it runs on the library's side stream right behind the
            # previous batch's covariance kernels, so no stream sits in a hardware queue waiting for an event
            self._complete(*self._pending, stream)
            self._pending = None
        ticket = self.ext.extract_batch_device(d_images.data_ptr(), self.fpr, buf.data_ptr(), stream.cuda_stream)
        if self.ext.async_cov:
            if self._pending is not None:
                # (no collective / torch collective: completing batch i orders `stream` or the torch communication stream
                # after its records — that wait has to come behind batch i + 1's work in stream order)
                self._complete(*self._pending, stream)
            self._pending = (ticket, k)
        else:
            self._complete(ticket, k, stream)
        return self.gathered

    def flush(self, stream):
        """Complete the batch still in flight (async mode) and order `stream` after the last gather."""
        if self._pending is not None:
            self._complete(*self._pending, stream)
            self._pending = None
        self.sync(stream)
        return self.gathered

    def _settle(self, stream):
        """Non-collective path: turn the pending ticket into an event on `stream` (see _complete)."""
        import torch

        if self._lazy_ticket is not None:
            if not stream.cuda_stream:   # the legacy default stream's handle is 0, which the C ABI reads as "the handle's own stream"
                if self._consumer is None:
                    self._consumer = torch.cuda.Stream()
                stream = self._consumer
            self.ext.wait_records(self._lazy_ticket, stream.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(stream)
            self._gathered_event, self._lazy_ticket = ev, None

    def sync(self, stream):
        """Make `stream` wait until `gathered` is complete."""
        self._settle(stream)
        if self._gathered_event is not None:
            stream.wait_event(self._gathered_event)

    def comm_ranks(self):
        """{"library": ranks RCCL reports for the library's communicator (ncclCommCount) or None when torch's collective is the
        one in use, "process_group": torch.distributed's world size} — what actually took part in the gather."""
        import torch.distributed as dist

        lib = None
        if self._native:
            lib = self.ext.comm_count()
        return {"library": lib, "process_group": dist.get_world_size() if dist.is_initialized() else 1,
                "backend": dist.get_backend() if dist.is_initialized() else None}

    def time_gather(self, iters=20):
        """The collective alone: `iters` all-gathers of the last completed batch's local records, bracketed by events on the
        stream the collective runs on.  Called by EVERY rank.  -> dict(ms per gather, bytes per rank, bus GB/s)."""
        import time as _time

        import torch
        import torch.distributed as dist

        if not self._collective:
            return {"ms": 0.0, "bytes_per_rank": 0, "iters": 0}
        k = (self._calls - 1) % len(self.local)
        local, out = self.local[k], self.all[self._gathers % 2]
        ticket = self.ext.last_ticket()
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def once():
            if self._native:
                self.ext.allgather_records(ticket, local.data_ptr(), out.data_ptr(), self.fpr)
            else:
                with torch.cuda.stream(self.comm):
                    if self._gather_fn is not None:
                        self._gather_fn(out, local)
                    else:
                        dist.all_gather_into_tensor(out, local)
        once()
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        t0 = _time.perf_counter()
        e0.record(self.comm)
        for _ in range(iters):
            once()
        e1.record(self.comm)
        e1.synchronize()
        torch.cuda.synchronize()
        wall = (_time.perf_counter() - t0) / iters * 1e3
        ms = e0.elapsed_time(e1) / iters
        self._gathers += 1                       # `out` now holds the re-gathered batch: it is the current result
        self.gathered, self._gathered_event = out, e1
        nb = local.numel()
        return {"ms": round(ms, 4), "wall_ms": round(wall, 4), "bytes_per_rank": int(nb), "iters": iters,
                "recv_GBps_per_rank": round((self.world - 1) * nb / (max(ms, 1e-6) * 1e-3) / 1e9, 2),
                "stream": "libspfe side stream (ncclAllGather)" if self._native else "torch communication stream"}

    def decode(self, frame):
        """Host copy + decode of global frame `frame` of the last completed batch."""
        import torch

        self._settle(torch.cuda.current_stream())
        if self._gathered_event is not None:
            self._gathered_event.synchronize()
        rb = self.rec_bytes
        return self.ext.view_record(self.gathered[frame * rb:(frame + 1) * rb].cpu().numpy())
