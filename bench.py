#!/usr/bin/env python3
"""bench.py — frames/sec of the SuperPoint extraction path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the ranks are one process per GPU under
torch.distributed.run — launched by the caller (the driver's form) or, when the environment carries no rendezvous
(RANK / WORLD_SIZE unset), BY THIS SCRIPT: it re-executes itself under `torch.distributed.run --nproc-per-node N`, so a
plain `python bench.py --gpus 8` can never run on one GPU and print `n_gpus: 1`.  An N > 1 run exits non-zero — and
prints no JSON line — when fewer than N devices are visible, when two ranks sit on the same device, or when the
library's RCCL communicator reports another rank count than N (SPFE_BENCH_BACKEND=gloo, the shared-GPU dry run of the
tests, lifts the device checks).  A step = one pass of the whole hot
path (u8 frames resident in HBM -> fixed-stride keypoint/descriptor records in
HBM, plus the RCCL all-gather of the records when N > 1) over a batch of
FRAMES_PER_GPU frames per GPU.  Prints ONE JSON line on rank 0.

Workload = BASELINE.json configs[1] scaled the way configs[2] shards it:
752x480 frames, num_features = 1000, f32, 8 independent frames per GPU per step
(64 frames on 8 GPUs); weak scaling.  The batch-1 latency of configs[1] is
reported beside it as `latency_batch1_ms`.

Order of the legs (N = 1): every GPU leg first, back to back — the headline, the per-stage table, the other
resolutions / dtypes north_star lists (640x480 and 1280x720 in f32, 752x480 and 1280x720 = configs[3] in bf16), batch-1
latency, the host boundary (f32 and bf16), matching, dust alignment, input staging, the tracker's front-end chain — then
the CPU legs (oracle = cpu_baseline + the self-checks, ATen-CPU).  N > 1: the headline, then what makes the line prove
itself: `parity_gathered` (a frame computed on ANOTHER rank, as it arrived through the all-gather, against the oracle),
`allgather_ms`, `rccl_ranks`, `host_alt` (each rank copies its own shard to its host instead of gathering; SURVEY.md
§8e), a barrier, and only then teardown on every rank.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_FRAME = {(480, 752): 61221703680, (480, 640): 52103577600, (720, 1280): 156310732800}   # SURVEY.md 8(d), exact


def flop_per_frame(H, W):
    """2 x sum over the twelve layers of (H / s)(W / s) Cin Cout k^2 (s = 1, 1, 2, 2, 4, 4, 8 ...); the three
    sizes SURVEY quotes are kept as the table above and checked against this formula."""
    layers = [(1, 1, 64, 3), (1, 64, 64, 3), (2, 64, 64, 3), (2, 64, 64, 3), (4, 64, 128, 3), (4, 128, 128, 3),
              (8, 128, 128, 3), (8, 128, 128, 3), (8, 128, 256, 3), (8, 256, 65, 1), (8, 128, 256, 3), (8, 256, 256, 1)]
    return 2 * sum((H // s) * (W // s) * cin * cout * k * k for s, cin, cout, k in layers)


assert all(flop_per_frame(h, w) == v for (h, w), v in FLOP_PER_FRAME.items())
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)


CLOCK_PROBE = os.path.join(ROOT, "tools", "microbench", "bin", "clock_probe")
_sustained = {}   # dtype -> {"tflops", "ghz"}: the last probe's figures (device of this rank)


def clock_probe(device):
    """Bare MFMA rate the part sustains per dtype, right now (tools/microbench/clock_probe --json: random operands, one
    wavefront per SIMD on every CU, ~50 ms per dtype after a warm-up launch of the same length; its own process on the same
    device).  The nominal-clock peaks (157.3 / 2500 TFLOP/s) assume 2.4 GHz; under dense MFMA the chip holds 1.9 - 2.2 GHz
    depending on the box and on the dtype, so every `frac` of the line is printed beside `frac_of_sustained`.  Returns the
    probe's dict, or {"error": ...}."""
    import subprocess
    if not os.path.exists(CLOCK_PROBE):
        return {"error": "tools/microbench/bin/clock_probe not built (__graft_entry__.build())"}
    try:
        r = subprocess.run([CLOCK_PROBE, "--json", str(device)], capture_output=True, text=True, timeout=60)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:   # noqa: BLE001 — a diagnostic leg must not cost the line
        return {"error": "%s: %s" % (type(e).__name__, e)}
    for k in ("f32", "bf16"):
        if k in d:
            _sustained[k] = d[k]
    return d


class SclkSampler:
    """Shader clock of the device during a timed region, read from the driver's sysfs node (pp_dpm_sclk: the line marked
    `*` is the level the SMU reports) every ~2 ms by a thread — min / avg / max MHz, or None where the node is absent."""

    def __init__(self, torch, device):
        import glob
        self.paths, self.samples, self._stop, self._th = [], [], False, None
        try:   # the card whose PCI function (…/0000:BB:DD.F) carries this device's bus number
            bus = int(getattr(torch.cuda.get_device_properties(device), "pci_bus_id"))
        except Exception:   # noqa: BLE001
            bus = None
        cands = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        if bus is not None:
            def bus_of(c):
                parts = os.path.basename(os.path.realpath(os.path.dirname(c))).split(":")
                try:
                    return int(parts[1], 16)
                except (IndexError, ValueError):
                    return None
            cands = [c for c in cands if bus_of(c) == bus] or cands
        self.paths = cands[:1]

    @staticmethod
    def _read(path):
        with open(path) as f:
            for ln in f:
                if "*" in ln:
                    return float(ln.split(":")[1].strip().lower().replace("mhz", "").replace("*", "").strip())
        return None

    def start(self):
        if not self.paths:
            return
        import threading

        def run():
            while not self._stop:
                try:
                    v = self._read(self.paths[0])
                    if v:
                        self.samples.append(v)
                except Exception:   # noqa: BLE001
                    return
                time.sleep(0.002)
        self._th = threading.Thread(target=run, daemon=True)
        self._th.start()

    def stop(self):
        self._stop = True
        if self._th:
            self._th.join(timeout=1.0)
        if not self.samples:
            return None
        s = self.samples
        return {"min": min(s), "avg": round(sum(s) / len(s), 1), "max": max(s), "samples": len(s), "source": self.paths[0]}


def bind_to_gpu_numa(torch, device):
    """N > 1: this rank's host threads onto the cores of its GPU's NUMA node (/sys/bus/pci/devices/<addr>/numa_node ->
    /sys/devices/system/node/nodeK/cpulist, intersected with the affinity the launcher left).  Eight ranks that all run on
    the launcher's cores submit their kernels across the socket interconnect and share one node's memory bandwidth for the
    pinned buffers; VERDICT r4 item 8.  Returns what was done (for the line's config), never raises."""
    try:
        pr = torch.cuda.get_device_properties(device)
        addr = "%04x:%02x:%02x.0" % (int(getattr(pr, "pci_domain_id", 0)), int(pr.pci_bus_id), int(pr.pci_device_id))
        with open("/sys/bus/pci/devices/%s/numa_node" % addr) as f:
            node = int(f.read().strip())
        if node < 0:
            return {"pci": addr, "numa_node": node, "bound": False, "why": "the platform reports no NUMA node for this device"}
        cpus = set()
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0) & cpus
        if not allowed:
            return {"pci": addr, "numa_node": node, "bound": False, "why": "none of the node's cores is in this process's affinity mask"}
        os.sched_setaffinity(0, allowed)
        return {"pci": addr, "numa_node": node, "bound": True, "cores": len(allowed)}
    except Exception as e:   # noqa: BLE001 — sysfs layouts differ; an unbound rank still measures
        return {"bound": False, "why": "%s: %s" % (type(e).__name__, e)}


def conv1b_flop(H, W):
    return 2 * H * W * 64 * 64 * 9


def path_tflops(ext, fps, H, W, B, precision=None):
    """Whole-path TFLOP/s = EXECUTED flops per frame x frames/s.  The descriptor head (convDb, 2 x 256 x 256 flops per coarse
    cell) runs on the cells the emitted keypoints' bilinear taps read only (libspfe's gathered head, on by default in f32
    mode and for bf16 frames of >= 10,000 cells): the rows it skips are not counted.  `dense_graph` is the reference's dense
    graph over the same time, for comparison with earlier rounds."""
    nominal = FLOP_PER_FRAME.get((H, W)) or flop_per_frame(H, W)
    C = (H // 8) * (W // 8)
    try:
        da = bool(ext.debug_read("da_gathered")[0])                  # (read first: reading db_total does not change it)
        frac = float(ext.debug_read("db_total")[0]) / float(B * C)   # the last call's list
        gathered = True
    except Exception:
        frac, gathered, da = 1.0, False, False
    executed = nominal - (1.0 - frac) * C * (2 * 256 * 256 + (2 * 9 * 128 * 256 if da else 0))
    out = {"whole_path_tflops": round(fps * executed / 1e12, 2),
           "whole_path_tflops_dense_graph": round(fps * nominal / 1e12, 2),
           "descriptor_head": {"convDb_gathered": gathered, "convDa_gathered": da, "cells_computed_frac": round(frac, 4)}}
    if precision:   # executed flops of the whole path against the nominal-clock MFMA peak and against what this box sustains
        peak = PEAK_BF16_MFMA_TFLOPS if precision == "bf16" else PEAK_F32_MFMA_TFLOPS
        sus = _sustained.get(precision)
        out["whole_path_frac_of_peak"] = round(fps * executed / 1e12 / peak, 4)
        out["whole_path_frac_of_sustained"] = round(fps * executed / 1e12 / sus["tflops"], 4) if sus else None
    return out


def file_sha16(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def stamped_traffic(profile_json, kernel_sources, H, W, B):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile, or None when the
    profile was taken on other kernel sources than the ones built now (stale) or another workload."""
    path = os.path.join(ROOT, "profiles", profile_json)
    try:
        d = json.load(open(path))
    except Exception:
        return None
    if tuple(d.get("workload_hwb", (480, 752, 8))) != (H, W, B):
        return None
    want = d.get("kernel_source_sha16") or {}
    for src in kernel_sources:
        if want.get(src) != file_sha16(os.path.join(ROOT, "sp_orb_slam_amd", "csrc", src)):
            return None
    return d.get("hbm_bytes_per_launch")


def traffic_of(precision, H, W, B):
    if precision == "bf16":
        srcs = ["conv_bf16_ws.hip", "conv1a_mfma.h"]
        return (stamped_traffic("conv1b_bf16_traffic.json", srcs, H, W, B) or
                stamped_traffic("conv1b_bf16_720p_traffic.json", srcs, H, W, B))
    return stamped_traffic("conv1b_traffic.json", ["conv_f32.hip"], H, W, B)


def run_timed(ext, sharded, d_img, stream, steps, warmup, world, dist, torch, sampler=None):
    """W untimed + K timed steps bracketed by barrier + synchronize; returns seconds (max over ranks).  `sampler`: an
    SclkSampler started when the timed region starts (a host thread reading a sysfs node: nothing on the device)."""
    for _ in range(warmup):
        sharded.step(d_img, stream)
    sharded.flush(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ext.stage_reset()
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    for _ in range(steps):
        sharded.step(d_img, stream)
    sharded.flush(stream)   # the last batch's covariance + gather are inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def conv1b_rows(ext, precision, H=0, B=0):
    """(rows per tile of the f32 conv1b instantiation the last call launched — 8 or 16, part of the kernel's name —, share of
    conv1b's work in that launch): large launches whose work list divides neither way run the first k tile rows of the batch
    as 16-row tiles and the rest as 8-row tiles in a second launch; the in-region events then bracket the 16-row KERNEL."""
    if precision != "f32":
        return 8, 1.0
    try:
        rows = int(ext.debug_read("conv1b_tile_rows")[0])
        k = int(ext.debug_read("conv1b_split_rows")[0])
    except Exception:
        return 8, 1.0
    if k > 0 and H and B:
        ty16 = (H + 15) // 16
        # (exact share in output rows: a frame's last tile row may be half a tile)
        f, r = divmod(k, ty16)
        return rows, (f * H + min(16 * r, H)) / float(B * H)
    return rows, 1.0


def roofline_of(precision, stages, H, W, B, traffic, tile=(8, 1.0)):
    tile_rows, share = tile
    t_conv1b = stages.get("conv1b", 0.0) * 1e-3
    bf16 = precision == "bf16"
    # bf16: the dominant kernel computes conv1a (9 taps, 1 -> 64 channels) as well, in its producer waves
    flop = (conv1b_flop(H, W) + (2 * H * W * 64 * 9 if bf16 else 0)) * share
    ach = (flop * B / t_conv1b / 1e12) if t_conv1b > 0 else None
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
    sus = _sustained.get("bf16" if bf16 else "f32")
    return {"bound": "mfma", "kernel": ("conv_bf16_ws_kernel<true,2> (conv1b with conv1a computed by its producer waves)" if bf16 else
                                        "conv_f32_kernel<1,64,3,16,4,1,%d,2,true,true> (conv1b, %d-row tiles)" % (tile_rows // 4, tile_rows)),
            "achieved": round(ach, 2) if ach else None, "peak": peak, "unit": "TFLOP/s",
            "frac": round(ach / peak, 4) if ach else None, "traffic": traffic,
            # the bare-MFMA rate this box sustained for this dtype in the probe nearest to this leg (clock_probe()): what the
            # nominal-clock `peak` shrinks to under load here, and the kernel's fraction of THAT
            "mfma_sustained_tflops": sus["tflops"] if sus else None, "mfma_sustained_ghz": sus["ghz"] if sus else None,
            "frac_of_sustained": round(ach / sus["tflops"], 4) if (ach and sus) else None,
            "kernel_ms": round(t_conv1b * 1e3, 4), "share_of_conv1b_in_this_launch": round(share, 4)}


def parity_of(rec, ref, bf16):
    """A device record against the oracle's extraction of the same frame -> (ok, detail)."""
    import numpy as np
    kp_ok = rec.K == ref["K"] and np.array_equal(rec.kp_xy, ref["kp_xy"]) and np.array_equal(rec.occ_grid, ref["occ_grid"])
    if bf16:
        a = {(int(x), int(y)) for x, y in rec.kp_xy}
        b = {(int(x), int(y)) for x, y in ref["kp_xy"]}
        idx = {(int(x), int(y)): i for i, (x, y) in enumerate(ref["kp_xy"])}
        pairs = [(i, idx[k]) for i, k in enumerate((int(x), int(y)) for x, y in rec.kp_xy) if k in idx]
        cos = [float(np.dot(rec.descriptors[i], ref["desc"][j])) for i, j in pairs]
        mabs = max((float(np.abs(rec.descriptors[i] - ref["desc"][j]).max()) for i, j in pairs), default=0.0)
        jac = len(a & b) / max(1, len(a | b))
        return bool(jac >= 0.87 and (not cos or min(cos) >= 0.999) and mabs <= 2e-2), {
            "rule": "bf16 mode vs the f32 oracle (SURVEY 8c; tests/golden/flip_report_bf16.json: 256 frames, Jaccard >= 0.890): keypoint-set "
                    "Jaccard >= 0.87, descriptors of common keypoints: cosine >= 0.999, max-abs <= 2e-2",
            "jaccard": round(jac, 4), "desc_cos_min": round(min(cos), 6) if cos else None, "desc_max_abs": round(mabs, 6)}
    desc_bits = kp_ok and np.array_equal(rec.descriptors.view(np.uint32), ref["desc"].view(np.uint32))
    cov_bits = kp_ok and np.array_equal(rec.cov2.view(np.uint32), ref["cov2"].view(np.uint32)) and \
        np.array_equal(rec.cov2_inv.view(np.uint32), ref["cov2_inv"].view(np.uint32))
    return bool(kp_ok and desc_bits and cov_bits), {
        "rule": "f32 mode vs the oracle: keypoints / occ_grid exact, descriptors and cov2 / cov2_inv bitwise",
        "keypoints_exact": bool(kp_ok), "desc_bitwise": bool(desc_bits), "cov2_bitwise": bool(cov_bits), "K": int(rec.K)}


def device_leg(ctx, precision, H, W, B, seed0, steps, warmup, what):
    """One more (precision, resolution) on its own handle: the same pipelined schedule, the same in-region event bracket
    around conv1b and the same timing rules as the headline.  Compact sub-object."""
    torch, parallel, synth, SPExtractor = ctx["torch"], ctx["parallel"], ctx["synth"], ctx["SPExtractor"]
    os.environ["SPFE_STAGE_TIMING"] = "2"
    ext = SPExtractor(ctx["nf"], H, W, ctx["blob"], max_batch=B, device=ctx["local"], with_heat=False,
                      async_cov=not ctx["sync_cov"], precision=precision)
    d = torch.from_numpy(synth.make_batch(seed0, B, H, W)).cuda()
    sh = parallel.ShardedExtractor(ext, 1, 0, B)
    smp = SclkSampler(torch, ctx["local"])
    dt = run_timed(ext, sh, d, ctx["stream"], steps, warmup, 1, ctx["dist"], torch, smp)
    sclk = smp.stop()
    st = ext.stage_times()
    r = sh.decode(0)
    ok = bool(0 < r.K <= ctx["nf"] + 1 and r.status == 0)
    fps = B * steps / dt
    clock_probe(ctx["local"])   # the rate the box sustains right behind this leg (-> roofline_of, path_tflops)
    out = {"what": "%s, %d timed steps after %d untimed" % (what, steps, warmup),
           "value": round(fps, 2), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 4), "dtype": precision,
           "roofline": roofline_of(precision, st, H, W, B, traffic_of(precision, H, W, B), conv1b_rows(ext, precision, H, B)),
           "sclk_mhz": sclk, "records_ok": ok}
    out.update(path_tflops(ext, fps, H, W, B, precision))
    ext.close()
    del d
    os.environ["SPFE_STAGE_TIMING"] = "0"
    return out


def host_path_leg(ctx, precision, H, W, B, frames, fps_device, steps):
    """The host boundary of operator() (sp_extractor.cpp:379-390 upload, :427-433 D2H): frames start in pageable host
    memory, results end as host views of the records — PCIe inclusive, never `value`.  Pipelined (spfe_submit_batch /
    spfe_collect_batch: pinned staging, H2D of batch i+1 and D2H of batch i-1 beside the compute of batch i) and
    synchronous (spfe_extract_batch)."""
    import numpy as np
    torch, SPExtractor, nf = ctx["torch"], ctx["SPExtractor"], ctx["nf"]
    exth = SPExtractor(nf, H, W, ctx["blob"], max_batch=B, device=ctx["local"], with_heat=False, precision=precision)
    himgs = [np.array(f) for f in frames[:B]]
    kh = max(100, steps)
    for _ in range(3):
        exth.extract_batch(himgs)
    t1 = time.perf_counter()
    for _ in range(kh):
        exth.extract_batch(himgs)
    dt_sync = time.perf_counter() - t1
    tk = [exth.submit_batch(himgs) for _ in range(2)]
    for _ in range(10):
        tk.append(exth.submit_batch(himgs))
        exth.collect_batch(tk.pop(0), copy=False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(kh):
        tk.append(exth.submit_batch(himgs))
        res = exth.collect_batch(tk.pop(0), copy=False)
    dt_pipe = time.perf_counter() - t1
    k_ok = all(0 < res[i].K <= nf + 1 and res[i].status == 0 for i in range(B))
    while tk:
        exth.collect_batch(tk.pop(0), copy=False)
    ext1h = SPExtractor(nf, H, W, ctx["blob"], max_batch=1, device=ctx["local"], with_heat=False, precision=precision)
    for _ in range(10):
        ext1h(himgs[0], None)
    t1 = time.perf_counter()
    for _ in range(200):
        ext1h(himgs[0], None)
    dt_one = (time.perf_counter() - t1) / 200
    out = {"what": "pageable host frames in -> host views of the records out (C ABI boundary, no heat maps), %dx%d %s, "
                   "%d frames per call, %d calls; pipelined = spfe_submit_batch/spfe_collect_batch, 3 in flight"
                   % (W, H, precision, B, kh),
           "fps": round(B * kh / dt_pipe, 2), "ms_per_call": round(dt_pipe / kh * 1e3, 4),
           "fps_synchronous": round(B * kh / dt_sync, 2), "ms_per_call_synchronous": round(dt_sync / kh * 1e3, 4),
           "frac_of_device_resident": round(B * kh / dt_pipe / fps_device, 4) if fps_device else None,
           "bytes_h2d": int(B * H * W), "bytes_d2h": int(B * exth.record_bytes()),
           "single_frame_operator_call_ms": round(dt_one * 1e3, 4), "records_ok": bool(k_ok)}
    exth.close()
    ext1h.close()
    return out


def dropin_leg(H, W, nf, blob, frame, lazy=False):
    """The drop-in call AS THE BACK-END MAKES IT: the compiled orbslam::SPExtractor : BaseExtractor with heat
    maps on (its default), through the base pointer + dynamic_cast, with the copies Frame::ExtractORB makes (frame.cpp:296-311:
    getCov2Inv(), dense_dust_.clone(), heat_.clone(), occ_grid_.copyTo()) — host frame in, cv::KeyPoint / cv::Mat out.  A C++
    program (tools/dropin/dropin_latency.cpp, built by __graft_entry__.build()); its own process, its own handle."""
    import subprocess
    import tempfile
    from sp_orb_slam_amd import weights
    exe = os.path.join(ROOT, "tools", "dropin", "bin", "dropin_latency")
    if not os.path.exists(exe):
        return {"error": "tools/dropin/bin/dropin_latency not built (__graft_entry__.build())"}
    with tempfile.TemporaryDirectory() as td:
        wpath, ipath = os.path.join(td, "w.spfw"), os.path.join(td, "im.raw")
        weights.save(wpath, blob)
        frame.tofile(ipath)
        try:
            r = subprocess.run([exe, wpath, ipath, str(H), str(W), str(nf), "300", "30"] + (["lazy"] if lazy else []),
                               capture_output=True, text=True, timeout=120)
        except Exception as e:
            return {"error": str(e)}
    if r.returncode != 0:
        return {"error": "rc %d: %s" % (r.returncode, r.stderr[-300:])}
    d = json.loads(r.stdout.strip().splitlines()[-1])
    d["what"] = ("orbslam::SPExtractor::operator() through BaseExtractor* with heat maps on + Frame::ExtractORB's copies "
                 "(frame.cpp:296-311), host frame in -> cv::KeyPoint / cv::Mat / Eigen out, %dx%d f32, PCIe inclusive "
                 "(record + %s D2H); the maps land in the members' own (page-locked) storage, the descriptor rows are copied "
                 "beside the covariance (spfe_set_map_buffers, spfe_extract_begin / _rows / _finish); operator_call_p50 = operator() "
                 "alone, the rest is Frame's clones" % (W, H, "heat_ only: the opt-in lazy form leaves heat_inv_ on the device until heatInv()" if lazy
                                        else "both heat maps: the reference's post-call state, the drop-in's default"))
    return d


def frontend_chain_leg(ctx, H, W, nframes, precision="f32"):
    """The tracker's per-frame front end on records that never leave HBM (the C5 substitute): per frame
    spfe_stage_batch_device (raw BGR -> gray) -> spfe_extract_batch_device -> spfe_track_dust_record_device
    (PoseOptimizationDust + patch-wise association at the alignment's projections, tracker_dust.cpp:92-172); only the pose
    block and the keypoint indices come back.  Returns (json object, what the CPU block needs for the parity check)."""
    import numpy as np
    from tools import track_scene as ts
    from sp_orb_slam_amd import weights
    from sp_orb_slam_amd.extractor import DUST_OUT_BYTES
    torch, SPExtractor, nf = ctx["torch"], ctx["SPExtractor"], ctx["nf"]
    blob = weights.synthetic(7, "trackable")
    world = ts.texture(21, *ts.world_size(H, W))
    ext = SPExtractor(nf, H, W, blob, max_batch=1, device=ctx["local"], with_heat=False, precision=precision)
    ext.set_staging(H, W, 3, False)
    rb = ext.record_bytes()
    stream = torch.cuda.Stream()
    raws = np.stack([np.repeat(ts.frame(world, k, H, W)[:, :, None], 3, 2) for k in range(nframes)])
    d_raw = torch.from_numpy(raws).cuda()                      # the camera's frames, resident (the staging boundary)
    d_gray = torch.zeros((1, H, W), dtype=torch.uint8, device="cuda")
    d_rec = torch.zeros(rb, dtype=torch.uint8, device="cuda")
    # pass 1 (untimed): the map points of frame k come from frame k-1's keypoints (host side of the tracker: local map)
    mp = [None]
    for k in range(nframes - 1):
        ext.stage_batch_device(d_raw[k].data_ptr(), 1, d_gray.data_ptr(), stream.cuda_stream)
        t = ext.extract_batch_device(d_gray.data_ptr(), 1, d_rec.data_ptr(), stream.cuda_stream)
        ext.wait_records(t, stream.cuda_stream)
        stream.synchronize()
        rec = ext.view_record(d_rec.cpu().numpy())
        pts, mpd, sel = ts.map_points(rec.kp_xy, rec.descriptors, k)
        mp.append((pts, mpd, rec.kp_xy[sel].copy()))
    NP = 192
    pts_all = np.zeros((nframes, NP, 3), np.float32)
    mpd_all = np.zeros((nframes, NP, 256), np.float32)
    T_all = np.zeros((nframes, 16), np.float32)
    n_all = np.zeros(nframes, np.int32)
    for k in range(1, nframes):
        n = min(len(mp[k][0]), NP)
        pts_all[k, :n], mpd_all[k, :n], n_all[k] = mp[k][0][:n], mp[k][1][:n], n
        T_all[k] = ts.start_pose(k).reshape(16)
    d_pts, d_mpd, d_T = torch.from_numpy(pts_all).cuda(), torch.from_numpy(mpd_all).cuda(), torch.from_numpy(T_all).cuda()
    d_out = torch.zeros(DUST_OUT_BYTES, dtype=torch.uint8, device="cuda")
    d_kp = torch.zeros(512, dtype=torch.int32, device="cuda")
    h_out = torch.zeros(DUST_OUT_BYTES, dtype=torch.uint8).pin_memory()
    h_kp = torch.zeros(512, dtype=torch.int32).pin_memory()
    keep = {}
    poses = {}
    lat, assoc, correct, inl = [], 0, 0, []
    torch.cuda.synchronize()
    for rep in range(2):                                        # rep 0 warms up
        lat = []
        for k in range(1, nframes):
            n = int(n_all[k])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ext.stage_batch_device(d_raw[k].data_ptr(), 1, d_gray.data_ptr(), stream.cuda_stream)
            t = ext.extract_batch_device(d_gray.data_ptr(), 1, d_rec.data_ptr(), stream.cuda_stream)
            ext.wait_records(t, stream.cuda_stream)
            ext.track_dust_record_device(d_rec.data_ptr(), d_pts[k].data_ptr(), d_mpd[k].data_ptr(), n, d_T[k].data_ptr(),
                                         d_out.data_ptr(), d_kp.data_ptr(), ts.FX, ts.FY, ts.CX, ts.CY, min_inliers=30,
                                         stream=stream.cuda_stream)
            with torch.cuda.stream(stream):
                h_out.copy_(d_out, non_blocking=True)
                h_kp.copy_(d_kp, non_blocking=True)
            stream.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
            if rep == 1:
                g = ext.decode_dust_out(h_out.numpy(), n)
                gk = h_kp.numpy()[:n].copy()
                m = gk >= 0
                inl.append(g["n_inlier"])
                poses[k] = g["Tcw"].astype(np.float64)
                if k in (1, nframes // 2, nframes - 1):        # frames the CPU block re-does with the oracle chain
                    keep[k] = dict(raw=raws[k], pts=pts_all[k, :n], mpd=mpd_all[k, :n], T0=T_all[k].reshape(4, 4), g=g, gk=gk,
                                   rec=d_rec.cpu().numpy())
                rec_xy = None
                if m.any():
                    rec_xy = ext.view_record(d_rec.cpu().numpy()).kp_xy
                    ox, oy = ts.offsets(k)
                    pox, poy = ts.offsets(k - 1)
                    d = rec_xy[gk[m]] - mp[k][2][:n][m]
                    assoc += int(m.sum())
                    correct += int(((d[:, 0] == -(ox - pox)) & (d[:, 1] == -(oy - poy))).sum())
    lat = sorted(lat)
    out = {"what": "tracker front end per frame, device resident: spfe_stage_batch_device (BGR %dx%d -> gray) -> "
                   "spfe_extract_batch_device -> spfe_track_dust_record_device (PoseOptimizationDust + patch association, "
                   "tracker_dust.cpp:92-172); D2H = pose block + keypoint indices only; %d frames, <= %d map points, "
                   "'trackable' synthetic weights (weights.py), %s" % (W, H, nframes - 1, NP, precision),
           "frontend_ms_per_frame": {"p50": round(lat[len(lat) // 2], 4), "p99": round(lat[int(len(lat) * 0.99) - 1], 4),
                                     "frames": len(lat)},
           "inliers_mean": round(float(np.mean(inl)), 1), "associations": assoc,
           "associations_consistent_with_camera_motion": round(correct / max(1, assoc), 4)}
    ext.close()
    # against the camera's true pose (tools/track_scene.pose): what the alignment recovers, whatever the precision
    err = [float(np.abs(poses[k][:3, 3] - ts.pose(*ts.offsets(k)).astype(np.float64)[:3, 3]).max()) for k in poses]
    out["pose_translation_err_vs_true"] = {"max": round(max(err), 6), "mean": round(float(np.mean(err)), 6)}
    return out, dict(keep=keep, blob=blob, H=H, W=W, poses=poses)


def frontend_chain_parity(fc, nf):
    """CPU block: the oracle chain on the kept frames of the front-end leg."""
    import numpy as np
    from oracle import oracle
    from tools import track_scene as ts
    from sp_orb_slam_amd import parallel
    lay = parallel.RecordLayout(fc["H"], fc["W"], nf)
    ok, detail = True, {}
    for k, v in fc["keep"].items():
        ref = oracle.extract(fc["blob"], oracle.stage_input(v["raw"], fc["H"], fc["W"]), nf)
        rec = lay.unpack(v["rec"])
        e_ok = rec["K"] == ref["K"] and np.array_equal(rec["kp_xy"], ref["kp_xy"]) and \
            np.array_equal(rec["desc"].view(np.uint32), ref["desc"].view(np.uint32)) and \
            np.array_equal(rec["dense_dust"].view(np.uint32), ref["dense_dust"].view(np.uint32))
        r = oracle.align_dust(ref["dense_dust"], v["pts"], v["T0"], ts.FX, ts.FY, ts.CX, ts.CY)
        rk = np.full(len(v["pts"]), -1, np.int32)
        if r["n_inlier"] >= 30:
            inl = np.flatnonzero(r["inlier"])
            rk[inl] = oracle.match_patches(v["mpd"][inl], r["uv"][inl], ref["occ_grid"], ref["desc"])
        a_ok = r["iterations"] == v["g"]["iterations"] and np.array_equal(r["inlier"], v["g"]["inlier"]) and \
            float(np.abs(r["Tcw"] - v["g"]["Tcw"]).max()) <= 1e-6
        m_ok = np.array_equal(rk, v["gk"])
        detail[str(k)] = {"extract_bitwise": bool(e_ok), "alignment": bool(a_ok), "associations_equal": bool(m_ok)}
        ok = ok and e_ok and a_ok and m_ok
    return bool(ok), detail


def multi_gpu_legs(ctx, args, ext, sharded, d_img, stream, frames_lo, B, H, W, emit_partial=None):
    """N > 1: what makes the line prove itself.  Collective calls are made by EVERY rank, in the same order."""
    import numpy as np
    torch, dist, parallel = ctx["torch"], ctx["dist"], ctx["parallel"]
    world, rank, nf = ctx["world"], ctx["rank"], ctx["nf"]
    res = {}
    # No leg below may cost the line.  These legs run collectives (and one re-makes communicators) on a topology this code has
    # never met before the driver's run, while the headline above is complete and checked at this point.  So a watchdog on every
    # rank holds a deadline for all the legs together (SPFE_LEGS_TIMEOUT, 300 s) and a nearer one for the comm-stream A/B leg
    # (SPFE_COMM_AB_TIMEOUT, 150 s): on expiry rank 0 prints the line as far as it got and every rank leaves through
    # os._exit(0).  A rank that FAILS inside a leg must not tear the others down either (torchrun would, and the line with
    # them): it reports on stderr and waits for its watchdog — the other ranks reach theirs inside the next collective.
    import threading
    import time
    wd = {"deadline": time.monotonic() + float(os.environ.get("SPFE_LEGS_TIMEOUT", "300")), "leg": "parity_gathered", "gather": None}

    def watchdog():
        while time.monotonic() < wd["deadline"]:
            time.sleep(0.25)
            if wd.get("done"):
                return
        if rank == 0 and emit_partial is not None:
            part = dict(res)
            if wd["gather"] is not None:
                part["allgather_ms"] = wd["gather"]["ms"]
                part["allgather"] = wd["gather"]
            part["legs_incomplete"] = "leg '%s' did not finish within its time limit; line printed without it and the legs behind it" % wd["leg"]
            if wd["leg"] == "comm_stream_ab":
                part["comm_stream_ab"] = {"error": "the comm-stream A/B leg did not finish within its time limit; line printed without it"}
            emit_partial(part)
        # Exit status.  By default 0 on every rank: the headline above is complete and checked, and a non-zero rank makes
        # torchrun tear the job down and report failure — a driver that then drops the line loses a valid measurement to a
        # diagnostic leg (`legs_incomplete` in the line and the stderr message are the trace).  SPFE_BENCH_STRICT_LEGS=1
        #  makes the outcome visible in the status instead: 3 on a rank that FAILED inside a leg, 4 where a leg
        # ran out of time — those ranks leave 2 s after the others, so that rank 0's line is out before torchrun reacts.
        if os.environ.get("SPFE_BENCH_STRICT_LEGS") == "1":
            time.sleep(2.0)
            os._exit(3 if wd.get("failed") else 4)
        os._exit(0)
    threading.Thread(target=watchdog, daemon=True).start()

    def inject(leg):   # (tests: SPFE_BENCH_FAIL_LEG=<leg>:<rank> makes that rank fail there)
        if os.environ.get("SPFE_BENCH_FAIL_LEG") == "%s:%d" % (leg, rank):
            raise RuntimeError("injected failure")

    def failed(leg, e):
        print("bench.py: rank %d: leg %s failed: %s: %s" % (rank, leg, type(e).__name__, e), file=sys.stderr, flush=True)
        wd["failed"] = leg
        threading.Event().wait()   # (until the watchdog ends the process)

    # (1) a frame computed on ANOTHER rank, as it arrived through the gather of the timed region
    if rank == 0 and not args.no_parity:
        try:
            from oracle import oracle
            from sp_orb_slam_amd import synth
            bf16 = args.precision == "bf16"
            oracle.set_num_threads(min(32, os.cpu_count() or 1))
            det = {}
            ok = True
            # the first frame of EVERY rank's shard (rank 0's is our own) and the last rank's last: each rank's contribution, as
            # it arrived through the gather, and both ends of the gathered buffer (world 8: 9 frames of the 64)
            for g in sorted({r * B for r in range(world)} | {world * B - 1}):
                ref = oracle.extract(ctx["blob"], synth.make_image(200 + g, H, W), nf)
                p, d = parity_of(sharded.decode(g), ref, bf16)
                det["frame_%d_from_rank_%d" % (g, g // B)] = d
                ok = ok and p
            res["parity_gathered"] = bool(ok)
            res["parity_gathered_detail"] = det
        except Exception as e:
            failed("parity_gathered", e)
    # (2) the collective alone: events on the stream it runs on
    wd["leg"] = "allgather"
    try:
        inject("allgather")
        res_g = sharded.time_gather(20)
    except Exception as e:
        failed("allgather", e)
    wd["gather"] = res_g
    # what took part in the HEADLINE's collective (read before the A/B leg below re-makes the communicator)
    try:
        ranks_headline = sharded.comm_ranks()
    except Exception as e:
        ranks_headline = {"library": "error: %s" % e, "process_group": world, "backend": None}
    if rank == 0:
        res["rccl_ranks"] = ranks_headline
    # (2b) where the collective runs: the library's side stream, right behind the covariance kernels of the batch it gathers
    # (default: the headline above) against a communication stream of its own that waits for the batch's event
    # (SPFE_COMM_OWN_STREAM=1).  With N > 1 the side-stream form puts batch i + 1's selection / descriptors / covariance behind
    # batch i's gather — i.e. behind the slowest rank; the own-stream form can land on the compute stream's hardware queue.
    # Both on THIS box, same handle, communicator re-made in between.
    ab = None
    if getattr(sharded, "_native", False) and not args.no_comm_ab:
        wd["leg"] = "comm_stream_ab"
        overall = wd["deadline"]
        wd["deadline"] = min(overall, time.monotonic() + float(os.environ.get("SPFE_COMM_AB_TIMEOUT", "150")))
        ab = {}
        k3, w3 = max(args.steps, 50), max(args.warmup, 5)
        prev = os.environ.get("SPFE_COMM_OWN_STREAM")
        try:
          for name, val in (("own_stream", "1"), ("side_stream", "0")):
              torch.cuda.synchronize()
              dist.barrier()
              ext.comm_destroy()
              os.environ["SPFE_COMM_OWN_STREAM"] = val
              sh3 = parallel.ShardedExtractor(ext, world, rank, B)
              if not getattr(sh3, "_native", False):
                  ab[name] = None
                  continue
              dt3 = run_timed(ext, sh3, d_img, stream, k3, w3, world, dist, torch)
              r3 = sh3.decode(world * B - 1)
              ab[name] = {"value": round(world * B * k3 / dt3, 2), "unit": "frames/s", "ms_per_step": round(dt3 / k3 * 1e3, 4),
                          "steps": k3, "records_ok": bool(0 < r3.K <= nf + 1 and r3.status == 0), "rccl_ranks": sh3.comm_ranks()["library"]}
              sharded = sh3   # (the handle's communicator is this one now: the legs below use it)
        except Exception as e:
            failed("comm_stream_ab", e)
        if prev is None:
            os.environ.pop("SPFE_COMM_OWN_STREAM", None)
        else:
            os.environ["SPFE_COMM_OWN_STREAM"] = prev
        wd["deadline"] = overall
        ab["what"] = ("the headline's schedule with ncclAllGather on a communication stream of its own (own_stream, "
                      "SPFE_COMM_OWN_STREAM=1) and on the library's side stream behind the batch's covariance (side_stream, the "
                      "default), %d timed steps each after %d untimed, same handle" % (k3, w3))
        if rank == 0:
            res["comm_stream_ab"] = ab
    wd["leg"] = "host_alt"
    # (3) the host-side alternative: no gather, every rank copies ITS shard to its own pinned host buffer
    nbytes = B * ext.record_bytes()
    try:
        inject("host_alt")
        pinned = [torch.zeros(nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
        cnt = [0]

        def d2h(out_unused, local):
            pinned[cnt[0] % 2].copy_(local, non_blocking=True)
            cnt[0] += 1
        sh2 = parallel.ShardedExtractor(ext, world, rank, B, gather_fn=d2h)
        k2 = max(args.steps, 50)
        dt2 = run_timed(ext, sh2, d_img, stream, k2, max(args.warmup, 5), world, dist, torch)
        own = ext.view_record(pinned[(cnt[0] - 1) % 2][:ext.record_bytes()].numpy())
    except Exception as e:
        failed("host_alt", e)
    wd["done"] = True
    if rank == 0:
        res["allgather_ms"] = res_g["ms"]
        res["allgather"] = res_g
        res["host_alt"] = {"what": "no collective: each rank D2H-copies its own %d records (%d bytes) to pinned host memory on a "
                                   "copy stream behind the batch's covariance (SURVEY.md 8e: the SLAM consumer is on the host), "
                                   "%d timed steps" % (B, nbytes, k2),
                           "value": round(world * B * k2 / dt2, 2), "unit": "frames/s",
                           "ms_per_step": round(dt2 / k2 * 1e3, 4), "records_ok": bool(0 < own.K <= nf + 1 and own.status == 0)}
        res["host_alt_ms"] = res["host_alt"]["ms_per_step"]
    return res, ranks_headline["library"]


# ---------------------------------------------------------------------------------------------------------------------
# The printed line.  The driver keeps the last 8 KB of the line it records (a 14 KB line lost BASELINE's own
# configs[3] leg to that window), so the DEFAULT line is compact — every leg reduced to its figures, no prose — and ENDS with
# a `configs` object that carries BASELINE.json's configurations in a fixed form; `--verbose-line` prints the full objects
# (the "what" strings, rules, details) instead, and the full line is always written to gpurun_out/bench_full.json when that
# directory exists.  tests/test_bench_line.py builds the line from canned leg objects and holds its length (<= 7500
# characters) and the position of `configs` (last key).
LINE_LIMIT = 7500
_DROP_KEYS = ("what", "rule", "stage_ms_note", "parity_detail", "source", "samples")


def _leg_summary(leg):
    """A device leg (device_leg() / the headline) -> the figures the `configs` object and the compact legs carry."""
    if not isinstance(leg, dict) or "value" not in leg:
        return leg
    rf = leg.get("roofline") or {}
    sc = leg.get("sclk_mhz") or {}
    out = {"value": leg.get("value"), "ms_per_step": leg.get("ms_per_step"), "kernel_ms": rf.get("kernel_ms"),
           "frac": rf.get("frac"), "frac_of_sustained": rf.get("frac_of_sustained"),
           "whole_path_frac_of_peak": leg.get("whole_path_frac_of_peak"),
           "whole_path_frac_of_sustained": leg.get("whole_path_frac_of_sustained"),
           "whole_path_tflops": leg.get("whole_path_tflops"), "whole_path_tflops_dense_graph": leg.get("whole_path_tflops_dense_graph"),
           "cells_computed_frac": (leg.get("descriptor_head") or {}).get("cells_computed_frac"),
           "mfma_sustained_tflops": rf.get("mfma_sustained_tflops"), "sclk_avg": sc.get("avg") if isinstance(sc, dict) else None}
    if "records_ok" in leg:
        out["records_ok"] = leg["records_ok"]
    return {k: v for k, v in out.items() if v is not None}


def _strip(v):
    """Drop prose and bulky details, recursively; round floats to 4 significant decimals where they are long."""
    if isinstance(v, dict):
        return {k: _strip(x) for k, x in v.items() if k not in _DROP_KEYS}
    if isinstance(v, list):
        return [_strip(x) for x in v]
    if isinstance(v, float):
        return float("%.6g" % v)
    return v


# device legs that live in `configs` only (compact line): BASELINE configs[3] and the other resolutions / dtypes north_star lists
_CONFIG_LEGS = (("bf16_1280x720_b8", "bf16_1280x720_b8 = configs[3]"), ("f32_640x480_b8", "f32_640x480_b8"),
                ("f32_1280x720_b8", "f32_1280x720_b8"), ("bf16_752x480_b8", "bf16_752x480_b8"))


def _configs_object(out):
    """BASELINE.json's configurations, one entry each, in a fixed form (the LAST key of the line)."""
    head = _leg_summary(dict(out, roofline=out.get("roofline"), sclk_mhz=out.get("sclk_mhz")))
    cfg = out.get("config") or {}
    hw = "%sx%s" % (cfg.get("width"), cfg.get("height"))
    lat = out.get("latency_batch1_ms") or {}
    drop = out.get("dropin_operator_call_ms") or {}
    dropl = out.get("dropin_operator_call_lazy_ms") or {}
    # order: what a reader of only the END of the line must still see comes last (the driver's record keeps a tail of the line:
    # 2,000 characters in BENCH_r05.json) — configs[1], configs[3] and the headline close the object
    c = {}
    for name, label in _CONFIG_LEGS:
        if name in out and "configs[3]" not in label:
            c[label] = _leg_summary(out[name])
    fc = {}
    for name in ("frontend_chain", "frontend_chain_bf16"):
        f = out.get(name)
        if isinstance(f, dict):
            fc[name] = {"ms_p50": (f.get("frontend_ms_per_frame") or {}).get("p50"), "inliers_mean": f.get("inliers_mean"),
                        "associations": f.get("associations"),
                        "consistent_with_camera_motion": f.get("associations_consistent_with_camera_motion"),
                        "final_pose_max_abs_diff_vs_f32": f.get("final_pose_max_abs_diff_vs_f32"),
                        "parity_vs_oracle_chain": f.get("parity_vs_oracle_chain")}
    if fc:
        c["configs[4] substitute (blocked: no dataset / weights / back-end)"] = fc
    c["headline = configs[2] per-GPU shard %s b%s %s x %s GPU" % (hw, cfg.get("frames_per_gpu"), out.get("dtype"), out.get("n_gpus"))] = head
    for name, label in _CONFIG_LEGS:
        if name in out and "configs[3]" in label:
            c[label] = _leg_summary(out[name])
    c["configs[1] %s b1 %s" % (hw, out.get("dtype"))] = {
        "latency_ms_p50": lat.get("p50"), "latency_ms_p99": lat.get("p99"), "calls": lat.get("calls"),
        "dropin_operator_call_ms_p50": drop.get("p50"), "dropin_operator_call_lazy_ms_p50": dropl.get("p50")}
    return c


def build_line(out, verbose=False):
    """The dict bench.py prints: `out` (every leg's full object) -> the compact default line, `configs` last.  Legs that are
    device legs shrink to _leg_summary(); prose goes; if the result still exceeds LINE_LIMIT the least important legs' objects
    are replaced by their headline figure, in a fixed order, until it fits (named in `line_trimmed`)."""
    if verbose:
        line = dict(out)
        line.pop("configs", None)
        line["configs"] = _configs_object(out)
        return line
    line = {}
    for k, v in out.items():
        if k in _DROP_KEYS or k == "configs" or k in dict(_CONFIG_LEGS):   # (those legs: in `configs`, once)
            continue
        if k == "roofline" and isinstance(v, dict):
            v = dict(v)
            if isinstance(v.get("kernel"), str):
                v["kernel"] = v["kernel"].split(" (")[0]
            line[k] = _strip(v)
        elif k == "config" and isinstance(v, dict):
            v = dict(v)
            if isinstance(v.get("workload"), str) and len(v["workload"]) > 150:
                v["workload"] = v["workload"][:147] + "..."
            line[k] = _strip(v)
        elif k == "mfma_sustained" and isinstance(v, dict):
            line[k] = {d: {"tflops": x.get("tflops"), "ghz": x.get("ghz")} for d, x in v.items() if isinstance(x, dict)} or _strip(v)
        elif k == "sclk_mhz" and isinstance(v, dict):
            line[k] = {"min": v.get("min"), "avg": v.get("avg"), "max": v.get("max")}
        elif isinstance(v, dict) and "roofline" in v and "value" in v:   # an exploratory device leg: its brief form
            line[k] = {x: y for x, y in _leg_summary(v).items() if x in ("value", "ms_per_step", "frac", "frac_of_sustained", "whole_path_frac_of_peak", "cells_computed_frac", "records_ok")}
        elif k == "cpu_baseline_aten" and isinstance(v, dict):
            line[k] = {x: v.get(x) for x in ("value", "unit", "cores", "host_cpus")}
        else:
            line[k] = _strip(v)
    cfgs = _configs_object(out)
    trimmed = []
    order = ["f32_3840x2160_b1", "bf16_3840x2160_b1", "latency_batch1_stage_ms", "stage_ms", "dense_descriptor_branch_b8",
             "f32_sparse_detector_b8", "bf16_sparse_detector_b8", "match_patches", "input_staging", "match_bruteforce",
             "dust_alignment", "host_path_bf16", "host_path", "frontend_chain", "frontend_chain_bf16", "cpu_baseline_aten",
             "latency_batch1_heat_maps_ms"]

    def assemble():
        full = dict(line)
        if trimmed:
            full["line_trimmed"] = trimmed
        full["configs"] = cfgs
        return full
    keep = ("config", "roofline", "cpu_baseline", "latency_batch1_ms", "dropin_operator_call_ms", "dropin_operator_call_lazy_ms")
    # (legs this list does not know: largest first, behind the known ones)
    rest = sorted((k for k, v in line.items() if isinstance(v, dict) and k not in order and k not in keep),
                  key=lambda k: -len(json.dumps(line[k])))
    for k in order + rest:
        if len(json.dumps(assemble())) <= LINE_LIMIT:
            break
        v = line.get(k)
        if isinstance(v, dict):
            small = {x: v[x] for x in ("value", "fps", "p50", "us_per_solve", "ms_per_batch", "us_per_call", "total") if x in v}
            if not small and isinstance(v.get("frontend_ms_per_frame"), dict):
                small = {"p50": v["frontend_ms_per_frame"].get("p50")}
            line[k] = small
            trimmed.append(k)
    return assemble()


def emit_line(out, args):
    """Print THE line (compact unless --verbose-line) and leave the full objects in gpurun_out/bench_full.json."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(d):
            with open(os.path.join(d, "bench_full.json"), "w") as f:
                f.write(json.dumps(build_line(out, verbose=True)) + "\n")
    except OSError:
        pass
    print(json.dumps(build_line(out, verbose=getattr(args, "verbose_line", False))), flush=True)


def self_launch(args):
    """`--gpus N` (N > 1) without a rendezvous in the environment: re-execute under torch.distributed.run, one rank per GPU,
    exactly as the driver would have (same flags, 127.0.0.1 rendezvous, a free port).  Never returns."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without RANK / WORLD_SIZE in the environment: launching %d ranks: %s"
          % (args.gpus, args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def die(msg, dist=None):
    """An N > 1 run that is not what --gpus asked for: no JSON line, a message, a non-zero exit on every rank."""
    print("bench.py: FATAL: " + msg, file=sys.stderr, flush=True)
    if dist is not None and dist.is_initialized():
        try:
            dist.destroy_process_group()
        except Exception:
            pass
    os._exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames-per-gpu", type=int, default=8)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=752)
    ap.add_argument("--num-features", type=int, default=1000)
    ap.add_argument("--detector", default="dense", choices=["dense", "sparse"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency probe")
    ap.add_argument("--sync-cov", action="store_true",
                    help="do not overlap the covariance stage of step i with the convolutions of step i+1")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-stage-table", action="store_true", help="skip the separate per-stage timing pass")
    ap.add_argument("--no-match", action="store_true", help="skip the matching, dust-alignment, staging and front-end legs (SURVEY 8f)")
    ap.add_argument("--no-bf16-leg", action="store_true", help="skip the other-resolution / other-dtype legs (incl. configs[3])")
    ap.add_argument("--no-uhd-leg", action="store_true", help="skip the 3840x2160 leg (4 GB of activations, ~6 s: synthetic frame + handle)")
    ap.add_argument("--no-aten", action="store_true", help="skip the ATen-CPU baseline")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive host-path legs")
    ap.add_argument("--no-parity", action="store_true", help="N > 1: skip the oracle check of the gathered records")
    ap.add_argument("--no-comm-ab", action="store_true", help="N > 1: skip the comm-stream A/B leg (gather on the library's side "
                    "stream vs on a stream of its own, SPFE_COMM_OWN_STREAM)")
    ap.add_argument("--latency-calls", type=int, default=1000)
    ap.add_argument("--verbose-line", action="store_true",
                    help="print every leg's full object (prose, rules, details): ~15 KB; the default line is compact (<= 7.5 KB, "
                         "`configs` last) because the driver keeps the last 8 KB of what it records")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16"],
                    help="f32: BASELINE configs[1]/[2] (bit-exact path, the headline); bf16: configs[3] (all twelve "
                         "convolutions, the two 1x1 heads included, as bf16 MFMA GEMMs with f32 accumulation; softmax, "
                         "NMS, descriptor sampling and covariance stay f32)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    have_rdzv = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not have_rdzv:
        self_launch(args)

    # timed region: HIP events around the dominant kernel only (two per step); the full per-stage table
    # comes from a separate short pass below (sixteen events per step cost 1-2 % of the throughput)
    os.environ.setdefault("SPFE_STAGE_TIMING", "2")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch  # device memory, streams, torch.distributed (RCCL); imported before libspfe
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:   # (a 1-rank environment with --gpus 8 is refused too: never a smaller run than asked for)
        die("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # SPFE_BENCH_BACKEND=gloo: dry run of the N > 1 flow on a box with fewer GPUs than ranks (ranks share
    # a device, the collective goes through gloo); the real runs use RCCL ("nccl"), one rank per GPU
    backend = os.environ.get("SPFE_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        die("no GPU visible")
    if backend != "nccl":
        local = local % max(1, ndev)
    elif world > 1 and ndev < world:
        die("--gpus %d but only %d device(s) visible: one rank per GPU is the contract (SPFE_BENCH_BACKEND=gloo is the "
            "shared-GPU dry run)" % (world, ndev))
    torch.cuda.set_device(local)
    numa = bind_to_gpu_numa(torch, local) if world > 1 else None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if backend == "nccl":
            # N ranks on N DISTINCT devices: every rank's (bus id, device uuid) through the process group
            pr = torch.cuda.get_device_properties(local)
            me = "%s/%s" % (getattr(pr, "pci_bus_id", local), getattr(pr, "uuid", local))
            ids = [None] * world
            dist.all_gather_object(ids, me)
            if len(set(ids)) != world:
                die("%d ranks but only %d distinct devices: %s" % (world, len(set(ids)), ids), dist)

    from sp_orb_slam_amd import parallel, synth, weights
    from sp_orb_slam_amd.extractor import SPExtractor

    H, W, nf, B = args.height, args.width, args.num_features, args.frames_per_gpu
    blob = weights.synthetic(7, args.detector)
    ext = SPExtractor(nf, H, W, blob, max_batch=B, device=local, with_heat=False, async_cov=not args.sync_cov,
                      precision=args.precision)
    rec_bytes = ext.record_bytes()
    # synthetic frames: seeds 200..; rank r owns global frames [r*B, (r+1)*B)
    lo, hi = parallel.shard_range(world * B, world, rank)
    frames = synth.make_batch(200 + lo, hi - lo, H, W)
    d_img = torch.from_numpy(frames).cuda()
    sharded = parallel.ShardedExtractor(ext, world, rank, B)
    stream = torch.cuda.Stream()   # compute stream (not the legacy default stream: no implicit barriers)
    torch.cuda.synchronize()
    ctx = dict(torch=torch, dist=dist, parallel=parallel, synth=synth, SPExtractor=SPExtractor, nf=nf, blob=blob, local=local,
               sync_cov=args.sync_cov, stream=stream, world=world, rank=rank)

    smp = SclkSampler(torch, local) if rank == 0 else None
    dt = run_timed(ext, sharded, d_img, stream, args.steps, args.warmup, world, dist, torch, smp)
    sclk = smp.stop() if smp else None
    stages = ext.stage_times()
    # the MFMA rate this box sustains per dtype, measured right behind the headline (rank 0's device; the other ranks idle
    # in the barrier of the legs below meanwhile)
    probe = clock_probe(local) if rank == 0 else None

    # sanity: the gathered records decode and carry the expected keypoint counts
    rec0 = sharded.decode(0)
    recl = sharded.decode(world * B - 1)
    assert 0 < rec0.K <= nf + 1 and 0 < recl.K <= nf + 1 and rec0.status == 0 and recl.status == 0

    bf16 = args.precision == "bf16"
    out = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        fps = world * B * args.steps / dt
        # dominant kernel: conv1b (43.5 % of the FLOPs), one launch covers B frames
        out = {
            "metric": "frames/sec SuperPoint extract (%dx%d, %s kpts)" % (W, H, "1k" if nf == 1000 else str(nf)),
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "%dx%d u8 frames, num_features=%d, %s, %d frames/GPU/step, "
                                   "%s synthetic detector weights; records all-gathered over RCCL when n_gpus>1; %s"
                                   % (W, H, nf,
                                      "all twelve convolutions (1x1 heads included) as bf16 MFMA GEMMs with f32 accumulation, "
                                      "f32 softmax / NMS / descriptor sampling / covariance"
                                      if bf16 else "f32 MFMA", B, args.detector,
                                      "covariance stage on the device, synchronous" if args.sync_cov else
                                      "covariance of step i overlapped with the convolutions of step i+1 (depth-2 pipeline)"),
                       "frames_per_gpu": B, "height": H, "width": W, "num_features": nf,
                       "parallelism": "dp%d" % world,
                       "rank0_numa_binding": numa,   # N > 1: every rank binds itself to its GPU's NUMA node (bind_to_gpu_numa)
                       "gather": ("none (1 GPU)" if world == 1 else
                                  "ncclAllGather inside libspfe (spfe_allgather_records)" if getattr(sharded, "_native", False)
                                  else "torch.distributed all_gather_into_tensor")},
            "roofline": roofline_of(args.precision, stages, H, W, B, traffic_of(args.precision, H, W, B), conv1b_rows(ext, args.precision, H, B)),
            # tools/microbench/clock_probe --json right behind the timed region: bare MFMA TFLOP/s and shader GHz per dtype on
            # random operands (nominal: 157.3 / 2500 TFLOP/s at 2.4 GHz) — a slow box and a regression read differently here
            "mfma_sustained": probe,
            "sclk_mhz": sclk,   # the SMU's reported shader clock level during the timed region (sysfs pp_dpm_sclk), or null
        }
        out.update(path_tflops(ext, fps, H, W, B, args.precision))

    if world > 1:
        # ---- N > 1: self-verification legs (every rank takes part), then ONE line, a barrier, and only then teardown
        os.environ["SPFE_STAGE_TIMING"] = "0"
        def emit_partial(part):
            line = dict(out)
            line.update(part)
            emit_line(line, args)
        res, lib_ranks = multi_gpu_legs(ctx, args, ext, sharded, d_img, stream, lo, B, H, W, emit_partial)
        if lib_ranks is not None and lib_ranks != world:   # (every rank checks its own communicator: all leave together)
            die("the library's RCCL communicator reports %s ranks, --gpus %d" % (lib_ranks, world), dist)
        if rank == 0:
            out.update(res)
            emit_line(out, args)
        torch.cuda.synchronize()
        dist.barrier()            # no rank destroys its communicator while another is still in a leg (or printing)
        ext.close()
        dist.barrier()
        dist.destroy_process_group()
        return

    # ---------------------------------------------------------------- N = 1: GPU legs, back to back
    fps = out["value"]
    cpu_todo = {}
    if not args.no_stage_table:
        # per-stage table: separate pass, same workload and schedule, all stages bracketed by events
        os.environ["SPFE_STAGE_TIMING"] = "1"
        ext_t = SPExtractor(nf, H, W, blob, max_batch=B, device=local, with_heat=False,
                            async_cov=not args.sync_cov, precision=args.precision)
        sh_t = parallel.ShardedExtractor(ext_t, 1, 0, B)
        for _ in range(2):
            sh_t.step(d_img, stream)
        sh_t.flush(stream)
        torch.cuda.synchronize()
        ext_t.stage_reset()
        for _ in range(8):
            sh_t.step(d_img, stream)
        sh_t.flush(stream)
        torch.cuda.synchronize()
        out["stage_ms"] = {k: round(v, 4) for k, v in ext_t.stage_times().items()}
        out["stage_ms_note"] = "separate 8-step pass with events around every stage (this rank only)"
        ext_t.close()
    os.environ["SPFE_STAGE_TIMING"] = "0"   # no events in the latency / matching / host-path legs
    if not args.no_bf16_leg:
        # What north_star lists beside the headline: "synthetic VGA / 752x480 / 1280x720 image batches", both dtypes.
        # >= 100 (f32) / >= 200 (bf16) timed steps: a 20-step region of a 0.6 - 1.1 ms bf16 step is 12 - 25 ms and reads 8 % low
        # — clocks and caches still settling (tools/microbench/run_steps_ab.sh: 6616 / 7200 / 7245 frames/s at 20 / 100 / 300
        # steps) — so these legs do not take the driver's --steps 20 literally; the headline does.
        legs = [("f32_640x480_b8", "f32", 480, 640, 400, 100, 10, "VGA 640x480, batch 8, f32 MFMA, 1 GPU"),
                ("f32_1280x720_b8", "f32", 720, 1280, 300, 100, 10, "1280x720, batch 8, f32 MFMA, 1 GPU"),
                ("bf16_752x480_b8", "bf16", 480, 752, 200, 200, 20, "752x480, batch 8, bf16 MFMA convolutions and heads (f32 accumulate), f32 softmax / NMS / descriptors / covariance, 1 GPU"),
                ("bf16_1280x720_b8", "bf16", 720, 1280, 300, 200, 20, "BASELINE configs[3]: 1280x720, batch 8, bf16 MFMA convolutions and heads (f32 accumulate), f32 softmax / NMS / descriptors / covariance, 1 GPU")]
        for name, prec, h2, w2, seed0, k2, wu2, what in legs:
            if (prec, h2, w2, 8) == (args.precision, H, W, B):
                continue
            out[name] = device_leg(ctx, prec, h2, w2, 8, seed0, max(k2, args.steps), max(wu2, args.warmup), what)
        # The largest frame the path takes (the reference accepts any multiple of 8, sp_extractor.cpp:70): 3840x2160, one frame
        # per step — 129,600 cells (select_huge_kernel), conv1a's output 2.12 GB a frame (the end of the convolutions' 32-bit
        # buffer offsets).  f32 only; exact against the oracle in tests/test_gpu_selection.py::test_full_extraction_2160p_matches_oracle
        if args.precision == "f32" and not args.no_uhd_leg:
            out["f32_3840x2160_b1"] = device_leg(ctx, "f32", 2160, 3840, 1, 500, 30, 3,
                                                 "3840x2160, one frame per step, f32 MFMA, 1 GPU (select_huge_kernel: 129,600 cells)")
            out["bf16_3840x2160_b1"] = device_leg(ctx, "bf16", 2160, 3840, 1, 500, 60, 5,
                                                  "3840x2160, one frame per step, bf16 MFMA convolutions and heads, 1 GPU (select_huge_kernel, two side chains)")
        # The headline workload with the SPARSE synthetic detector (~1-2 k candidates per frame in isolated peaks — what a trained
        # SuperPoint produces — instead of a candidate in every cell): the headline's cells_computed_frac, selection and
        # covariance costs are properties of the dense detector; this is the other end
        if args.detector == "dense":
            blob_keep = ctx["blob"]
            ctx["blob"] = weights.synthetic(7, "sparse")
            try:
                out["%s_sparse_detector_b%d" % (args.precision, B)] = device_leg(
                    ctx, args.precision, H, W, B, 200, max(100, args.steps), max(10, args.warmup),
                    "the headline workload (%dx%d, %s, batch %d) with the sparse synthetic detector weights" % (W, H, args.precision, B))
            finally:
                ctx["blob"] = blob_keep
        # The headline workload with the reference's dense graph FULLY executed (SPFE_SPARSE_DB=0: convDa / convDb over the whole
        # coarse map, as sp_extractor.cpp:99-100 runs them), for the reader who wants that number: the records are the same
        # bits either way (tests/test_gpu_sparse_db.py); the default path computes the rows the keypoints read
        prev_env = os.environ.get("SPFE_SPARSE_DB")
        os.environ["SPFE_SPARSE_DB"] = "0"
        try:
            out["dense_descriptor_branch_b8"] = device_leg(
                ctx, args.precision, H, W, B, 200, max(100, args.steps), max(10, args.warmup),
                "the headline workload (%dx%d, %s, batch %d) with convDa / convDb computed on every cell (SPFE_SPARSE_DB=0)"
                % (W, H, args.precision, B))
        finally:
            if prev_env is None:
                os.environ.pop("SPFE_SPARSE_DB", None)
            else:
                os.environ["SPFE_SPARSE_DB"] = prev_env
    # batch-1 latency (configs[1] as written: one frame per call)
    if not args.no_latency:
        # the split of one single-frame call by stage (events around every stage of the launch stream; this serialises the
        # descriptor branch behind the covariance chain, so `total` reads higher than the p50 below)
        os.environ["SPFE_STAGE_TIMING"] = "1"
        ext1 = SPExtractor(nf, H, W, blob, max_batch=1, device=local, with_heat=False, precision=args.precision)
        d1 = d_img[:1].contiguous()
        r1 = torch.zeros(rec_bytes, dtype=torch.uint8, device="cuda")
        for i in range(60):
            if i == 10:
                ext1.stage_reset()
            ext1.extract_batch_device(d1.data_ptr(), 1, r1.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
        out["latency_batch1_stage_ms"] = {k: round(v, 4) for k, v in ext1.stage_times().items()}
        ext1.close()
        os.environ["SPFE_STAGE_TIMING"] = "0"
        ext1 = SPExtractor(nf, H, W, blob, max_batch=1, device=local, with_heat=False, precision=args.precision)
        d1 = d_img[:1].contiguous()
        r1 = torch.zeros(rec_bytes, dtype=torch.uint8, device="cuda")
        lat = []
        for i in range(args.latency_calls + 50):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ext1.extract_batch_device(d1.data_ptr(), 1, r1.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t1) * 1e3)
        lat = sorted(lat[50:])   # 1000 timed calls after 50 warm-up calls
        out["latency_batch1_ms"] = {"p50": round(lat[len(lat) // 2], 4), "p99": round(lat[int(len(lat) * 0.99) - 1], 4),
                                    "calls": len(lat)}
        ext1.close()
        # the same call with the heat maps ON (the drop-in adaptor's default: sp_extractor.cpp:461-474 always fills heat_ /
        # heat_inv_, Frame clones heat_, frame.cpp:304), still device resident
        ext1 = SPExtractor(nf, H, W, blob, max_batch=1, device=local, with_heat=True, precision=args.precision)
        lat = []
        for i in range(250):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ext1.extract_batch_device(d1.data_ptr(), 1, r1.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t1) * 1e3)
        lat = sorted(lat[50:])
        out["latency_batch1_heat_maps_ms"] = {"p50": round(lat[len(lat) // 2], 4), "p99": round(lat[int(len(lat) * 0.99) - 1], 4),
                                              "calls": len(lat)}
        ext1.close()

    if not args.no_host_path:
        out["host_path"] = host_path_leg(ctx, args.precision, H, W, B, frames, fps, args.steps)
        out["dropin_operator_call_ms"] = dropin_leg(H, W, nf, blob, frames[0]) if args.precision == "f32" else None
        out["dropin_operator_call_lazy_ms"] = dropin_leg(H, W, nf, blob, frames[0], lazy=True) if args.precision == "f32" else None
        if not args.no_bf16_leg and not (bf16 and (H, W) == (720, 1280)):
            fr3 = synth.make_batch(300, 8, 720, 1280)
            dev3 = (out.get("bf16_1280x720_b8") or {}).get("value")
            out["host_path_bf16"] = host_path_leg(ctx, "bf16", 720, 1280, 8, fr3, dev3, args.steps)

    if not args.no_match:
        # SURVEY.md §8(f) rank 1 (outside the timed region): match this step's B frames against
        # the same frames shifted by one cell, records resident in HBM, cross-check on.
        extm = SPExtractor(nf, H, W, blob, max_batch=B, device=local, with_heat=False,
                           precision=args.precision)
        d_img2 = torch.roll(d_img, (8, 16), (1, 2)).contiguous()
        ra = torch.zeros(B * rec_bytes, dtype=torch.uint8, device="cuda")
        rb2 = torch.zeros(B * rec_bytes, dtype=torch.uint8, device="cuda")
        mo = torch.zeros(B * extm.match_out_bytes(), dtype=torch.uint8, device="cuda")
        mstream = torch.cuda.Stream()   # an explicit stream: a NULL stream argument means "the handle's own"
        torch.cuda.synchronize()
        extm.extract_batch_device(d_img.data_ptr(), B, ra.data_ptr(), mstream.cuda_stream)
        extm.extract_batch_device(d_img2.data_ptr(), B, rb2.data_ptr(), mstream.cuda_stream)
        for _ in range(3):
            extm.match_records_device(rb2.data_ptr(), ra.data_ptr(), B, mo.data_ptr(), True, mstream.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nit = 30
        e0.record(mstream)
        for _ in range(nit):
            extm.match_records_device(rb2.data_ptr(), ra.data_ptr(), B, mo.data_ptr(), True, mstream.cuda_stream)
        e1.record(mstream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / nit
        ka = [extm.view_record(ra[i * rec_bytes:(i + 1) * rec_bytes].cpu().numpy()) for i in range(B)]
        kb = [extm.view_record(rb2[i * rec_bytes:(i + 1) * rec_bytes].cpu().numpy()) for i in range(B)]
        pair_elems = sum(a.K * b.K for a, b in zip(ka, kb)) * 256
        mb = extm.match_out_bytes()
        idx0, _ = extm.decode_match_out(mo[:mb].cpu().numpy(), kb[0].K)
        out["match_bruteforce"] = {
            "what": "cv::BFMatcher(NORM_L2, crossCheck=true) rule, %d frame pairs per launch, K~%d x %d, "
                    "records in HBM" % (B, kb[0].K, ka[0].K),
            "ms_per_batch": round(ms, 4), "pairs_per_s": round(B / ms * 1e3, 1),
            # 1 subtract + 1 fma per descriptor element pair, on the f32 VALU
            "valu_tflops": round(pair_elems * 3 / (ms * 1e-3) / 1e12, 2),
            "matched_frac_pair0": round(float((idx0 >= 0).mean()), 3)}
        cpu_todo["match"] = (kb[0].descriptors.copy(), ka[0].descriptors.copy())
        # patch-wise association of 200 projected map points against frame 0's resident record
        # (tracker_dust.cpp:113-172; mps_for_track holds 150-200 points)
        f0 = ka[0]
        rngp = np.random.default_rng(3)
        kk = rngp.integers(0, f0.K, 200)
        mpd = f0.descriptors[kk] + np.float32(0.3) * (rngp.standard_normal((200, 256)).astype(np.float32) / 16)
        mpuv = np.stack([f0.kp_xy[kk, 0] // 8 + 0.5 - rngp.integers(0, 2, 200), f0.kp_xy[kk, 1] // 8 + 0.5 - rngp.integers(0, 2, 200)], 1)
        d_mpd, d_mpuv = torch.from_numpy(mpd.astype(np.float32)).cuda(), torch.from_numpy(mpuv.astype(np.float32)).cuda()
        d_pidx = torch.zeros(200, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for _ in range(3):
            extm.match_patches_record_device(d_mpd.data_ptr(), d_mpuv.data_ptr(), 200, ra.data_ptr(), d_pidx.data_ptr(),
                                             0.75, mstream.cuda_stream)
        e0.record(mstream)
        for _ in range(nit):
            extm.match_patches_record_device(d_mpd.data_ptr(), d_mpuv.data_ptr(), 200, ra.data_ptr(), d_pidx.data_ptr(),
                                             0.75, mstream.cuda_stream)
        e1.record(mstream)
        torch.cuda.synchronize()
        out["match_patches"] = {"what": "tracker_dust.cpp:113-172 association, 200 map points vs one resident record",
                                "us_per_call": round(e0.elapsed_time(e1) / nit * 1e3, 2),
                                "matched": int((d_pidx.cpu().numpy() >= 0).sum())}
        extm.close()

        # SURVEY.md §8(f) rank 3 (outside the timed region): direct dust alignment, 160 map points against the
        # dense_dust of frame 0's resident record (optimizer_dust.cpp:170-294: 40 LM iterations)
        from tools import dust_scene
        from sp_orb_slam_amd.extractor import DUST_MAX_POINTS, DUST_OUT_BYTES
        extd = SPExtractor(nf, H, W, blob, max_batch=1, device=local, with_heat=False)
        dsc = dust_scene.make_scene(0, H=H, W=W, n_points=160, cx=W / 2 - 8.8, cy=H / 2 + 8.4)
        gz = None
        if (H, W) == (480, 752):
            # the scene of tests/golden/dust_std0.npz: its expected pose / flags come from an INDEPENDENT f64 numpy / scipy
            # statement of the optimisation (tests/golden/make_golden_dust.py; no code shared with the kernel or the oracle)
            try:
                gz = np.load(os.path.join(ROOT, "tests", "golden", "dust_std0.npz"))
                fxg, fyg, cxg, cyg = (np.float32(v) for v in gz["intr"])
                dsc = dict(dust=gz["dust"], pts=gz["pts"], Tcw_init=gz["Tcw_init"], fx=fxg, fy=fyg, cx=cxg, cy=cyg)
            except Exception:
                gz = None
        d_rec = torch.zeros(rec_bytes, dtype=torch.uint8, device="cuda")
        dstream = torch.cuda.Stream()
        extd.extract_batch_device(d_img.data_ptr(), 1, d_rec.data_ptr(), dstream.cuda_stream)
        # a scene-shaped dust map in the record, so that the solve does real work (synthetic weights give a flat one)
        lay = parallel.RecordLayout(H, W, nf)
        d_rec[lay.off_dd:lay.off_dd + dsc["dust"].size * 4] = torch.from_numpy(dsc["dust"].reshape(-1).view(np.uint8)).cuda()
        d_pts, d_T = torch.from_numpy(dsc["pts"]).cuda(), torch.from_numpy(dsc["Tcw_init"].reshape(16)).cuda()
        d_do = torch.zeros(DUST_OUT_BYTES, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()

        def dust_once():
            extd.align_dust_record_device(d_rec.data_ptr(), d_pts.data_ptr(), 160, d_T.data_ptr(), d_do.data_ptr(),
                                          dsc["fx"], dsc["fy"], dsc["cx"], dsc["cy"], stream=dstream.cuda_stream)
        for _ in range(3):
            dust_once()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(dstream)
        for _ in range(20):
            dust_once()
        e1.record(dstream)
        torch.cuda.synchronize()
        gd = extd.decode_dust_out(d_do.cpu().numpy(), 160)
        out["dust_alignment"] = {"what": "Optimizer::PoseOptimizationDust: 160 map points vs the dense_dust of a resident "
                                         "record, Huber 0.9, <= 40 LM iterations, one workgroup, f64",
                                 "us_per_solve": round(e0.elapsed_time(e1) / 20 * 1e3, 1), "iterations": gd["iterations"],
                                 "n_inlier": gd["n_inlier"]}
        if gz is not None:
            out["dust_alignment"]["vs_independent_fixture"] = {
                "fixture": "tests/golden/dust_std0.npz",
                "pose_max_abs_diff": float(np.abs(gd["Tcw"].astype(np.float64) - gz["pose64"]).max()),
                "iterations_equal": bool(gd["iterations"] == int(gz["iterations"])),
                "inlier_flags_equal": bool(np.array_equal(gd["inlier"], gz["inlier"]))}
        # the batch path's form: one workgroup per frame, 64 independent solves (each its own record, points and start
        # pose) in one launch — a solve is a latency chain, so they take about as long as one
        NB = 64
        d_recs = d_rec.repeat(NB, 1).contiguous()
        pts_b = np.zeros((NB, DUST_MAX_POINTS, 3), np.float32)
        T_b = np.zeros((NB, 16), np.float32)
        for f in range(NB):
            scf = dsc if f == 0 else dust_scene.make_scene(f, H=H, W=W, n_points=160, cx=W / 2 - 8.8, cy=H / 2 + 8.4)  # frame 0 = the single solve's scene
            pts_b[f, :160] = scf["pts"]
            T_b[f] = scf["Tcw_init"].reshape(16)
            d_recs[f, lay.off_dd:lay.off_dd + scf["dust"].size * 4] = torch.from_numpy(scf["dust"].reshape(-1).view(np.uint8)).cuda()
        d_pb, d_Tb = torch.from_numpy(pts_b).cuda(), torch.from_numpy(T_b).cuda()
        d_nb = torch.full((NB,), 160, dtype=torch.int32, device="cuda")
        d_ob = torch.zeros((NB, DUST_OUT_BYTES), dtype=torch.uint8, device="cuda")

        def dust_batch():
            extd.align_dust_batch_device(d_recs.data_ptr(), NB, d_pb.data_ptr(), d_nb.data_ptr(), d_Tb.data_ptr(), d_ob.data_ptr(),
                                         dsc["fx"], dsc["fy"], dsc["cx"], dsc["cy"], stream=dstream.cuda_stream)
        for _ in range(2):
            dust_batch()
        e0.record(dstream)
        for _ in range(10):
            dust_batch()
        e1.record(dstream)
        torch.cuda.synchronize()
        ob = d_ob.cpu().numpy()
        out["dust_alignment"]["batch64_us_per_launch"] = round(e0.elapsed_time(e1) / 10 * 1e3, 1)
        out["dust_alignment"]["batch64_us_per_solve"] = round(e0.elapsed_time(e1) / 10 / NB * 1e3, 2)
        out["dust_alignment"]["batch64_frame0_equals_single"] = bool(np.array_equal(ob[0][:64], d_do.cpu().numpy()[:64]))
        cpu_todo["dust"] = (dsc, gd)
        extd.close()

        # SURVEY.md §8(f) rank 2 (outside the timed region): staging kernel on B raw BGR frames
        exts = SPExtractor(nf, H, W, blob, max_batch=B, device=local, with_heat=False)
        vv, uu = np.mgrid[0:H, 0:W].astype(np.float64)
        r2 = ((uu - W / 2) / W) ** 2 + ((vv - H / 2) / W) ** 2
        mx = (uu + (uu - W / 2) * (-0.28 * r2)).astype(np.float32)
        my = (vv + (vv - H / 2) * (-0.28 * r2)).astype(np.float32)
        exts.set_staging(H, W, 3, False, mx, my)
        d_raw = d_img[:, :, :, None].expand(B, H, W, 3).contiguous()
        d_gray = torch.zeros((B, H, W), dtype=torch.uint8, device="cuda")
        sstream = torch.cuda.Stream()
        torch.cuda.synchronize()
        for _ in range(3):
            exts.stage_batch_device(d_raw.data_ptr(), B, d_gray.data_ptr(), sstream.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(sstream)
        for _ in range(30):
            exts.stage_batch_device(d_raw.data_ptr(), B, d_gray.data_ptr(), sstream.cuda_stream)
        e1.record(sstream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        # algorithmic bytes: 4 taps x 3 ch gathered (hits L2: ~1 raw frame) + 2 maps + 1 gray out
        alg = B * H * W * (3 + 8 + 1)
        out["input_staging"] = {"what": "cv::remap(INTER_LINEAR) + crop + BGR2GRAY, %d frames %dx%dx3 per launch"
                                        % (B, W, H), "ms_per_batch": round(ms, 4),
                                "hbm_GBps_algorithmic": round(alg / (ms * 1e-3) / 1e9, 1)}
        exts.close()
        del d_raw

        # the tracker's front end chained on resident records (the C5 substitute)
        out["frontend_chain"], cpu_todo["frontend"] = frontend_chain_leg(ctx, H, W, 101)
        # ... and the same sequence with the bf16 convolutions: the bf16 keypoint sets differ from the f32
        # ones by 6 - 11 % (Jaccard 0.89 - 0.94) — what that does to what the tracker consumes: inliers of the alignment,
        # associations, their consistency with the known camera motion, and the recovered pose against the f32 chain's
        if args.precision == "f32":
            fcb, keepb = frontend_chain_leg(ctx, H, W, 101, "bf16")
            pf, pb = cpu_todo["frontend"]["poses"], keepb["poses"]
            dif = [float(np.abs(pf[k] - pb[k]).max()) for k in pf if k in pb]
            fcb["pose_max_abs_diff_vs_f32"] = {"max": round(max(dif), 6), "mean": round(float(np.mean(dif)), 6), "frames": len(dif)}
            fcb["final_pose_max_abs_diff_vs_f32"] = round(float(np.abs(pf[max(pf)] - pb[max(pb)]).max()), 6)
            fcb["consistency_vs_f32"] = round(fcb["associations_consistent_with_camera_motion"] -
                                              out["frontend_chain"]["associations_consistent_with_camera_motion"], 4)
            out["frontend_chain_bf16"] = fcb

    # ---------------------------------------------------------------- CPU legs (the oracle is the checker / the baseline)
    if not args.no_cpu_baseline:
        # CPU baseline: the C oracle (a port of the path; the reference has no CPU
        # path and cannot be built here) on this host's cores, bounded sample.
        from oracle import oracle
        # the port's loops scale to a few dozen threads (tools/cpu_baseline.py sweeps 1..all: the
        # best team on the 256-thread GPU-box host is 32), so that is the team it gets
        nthr = oracle.set_num_threads(min(32, os.cpu_count() or 1))
        ref0 = oracle.extract(blob, frames[0], nf)   # warm-up (thread team, caches) + the parity check below
        done, t1 = 0, time.perf_counter()
        while True:
            oracle.extract(blob, frames[done % B], nf)
            done += 1
            el = time.perf_counter() - t1
            if el >= args.cpu_seconds or done >= 512:
                break
        out["cpu_baseline"] = {"value": round(done / el, 3), "unit": "frames/s", "cores": nthr,
                               "kind": "port",
                               "sample": "%d frames of the same %dx%d workload through oracle/spfe_oracle.c "
                                         "(OpenMP, %d threads), %.1f s" % (done, W, H, nthr, el)}
        # self-check of the timed workload: frame 0 of the LAST timed batch (pipelined, device
        # resident) against the oracle's extraction of the same frame
        out["parity_frame0"], out["parity_frame0_detail"] = parity_of(rec0, ref0, bf16)
        if "match" in cpu_todo:
            t1 = time.perf_counter()
            oracle.match_bruteforce(cpu_todo["match"][0], cpu_todo["match"][1], True)
            out["match_bruteforce"]["cpu_oracle_ms_per_pair"] = round((time.perf_counter() - t1) * 1e3, 2)
        if "dust" in cpu_todo:
            dsc, gd = cpu_todo["dust"]
            t1 = time.perf_counter()
            rd = oracle.align_dust(dsc["dust"], dsc["pts"], dsc["Tcw_init"], dsc["fx"], dsc["fy"], dsc["cx"], dsc["cy"])
            out["dust_alignment"]["cpu_oracle_us_per_solve"] = round((time.perf_counter() - t1) * 1e6, 1)
            out["dust_alignment"]["pose_max_abs_diff_vs_oracle"] = float(np.abs(rd["Tcw"] - gd["Tcw"]).max())
        if "frontend" in cpu_todo:
            ok, det = frontend_chain_parity(cpu_todo["frontend"], nf)
            out["frontend_chain"]["parity_vs_oracle_chain"] = ok
            out["frontend_chain"]["parity_detail"] = det
        if not args.no_aten:
            # north_star: "the reference's CPU libtorch path timed on the same box's host cores".  The
            # reference has no CPU path (CUDA hard-wired); libtorch-CPU == ATen-CPU, and torch is here
            # for device plumbing, so the builder's statement of SPFrontend::forward's op sequence
            # (tools/aten_path.py) is timed on the best of a thread sweep.  Network + detector tail +
            # descriptor sampling only (the host glue nms / computeCovariance is not ATen code).
            from tools import aten_path
            named = weights.to_named_tensors(blob)
            ncpu = os.cpu_count() or 1
            sweep, best = {}, None
            for thr in [t for t in (8, 16, 32, 64, 128) if t <= ncpu] or [ncpu]:
                fps_a, n_a, el_a = aten_path.time_forward(named, list(frames), thr, seconds=2.5, max_frames=48)
                sweep[str(thr)] = round(fps_a, 2)
                if best is None or fps_a > best[0]:
                    best = (fps_a, thr, n_a, el_a)
            out["cpu_baseline_aten"] = {
                "value": round(best[0], 3), "unit": "frames/s", "cores": best[1],
                "kind": "aten-cpu op sequence (torch %s, MKL-DNN), network + tail + descriptor sampling" % torch.__version__,
                "sample": "%d frames of the same %dx%d workload, %.1f s, best of the thread sweep" % (best[2], W, H, best[3]),
                "thread_sweep_fps": sweep, "host_cpus": ncpu}
    emit_line(out, args)
    ext.close()


if __name__ == "__main__":
    main()
