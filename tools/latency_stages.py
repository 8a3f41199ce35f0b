#!/usr/bin/env python3
"""Batch-1 (configs[1] as written) split by stage: one 752x480 frame per spfe_extract_batch_device call, synchronous.
Two passes: (1) wall-clock p50 over N calls without events, (2) SPFE_STAGE_TIMING=1 (events around every stage on the
launch stream; `post_side` = the side chain serialised).  usage: tools/latency_stages.py [--precision f32|bf16] [--height H --width W]
Under rocprofv3 --kernel-trace --stats the same script gives the per-kernel durations of a single-frame call."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=752)
    ap.add_argument("--calls", type=int, default=300)
    ap.add_argument("--detector", default="dense")
    ap.add_argument("--heat", action="store_true")
    a = ap.parse_args()
    import torch
    from sp_orb_slam_amd import synth, weights
    from sp_orb_slam_amd.extractor import SPExtractor
    H, W = a.height, a.width
    blob = weights.synthetic(7, a.detector)
    d1 = torch.from_numpy(synth.make_batch(200, 1, H, W)).cuda()
    stream = torch.cuda.Stream()
    out = {}
    for mode in ("0", "1"):
        os.environ["SPFE_STAGE_TIMING"] = mode
        ext = SPExtractor(1000, H, W, blob, max_batch=1, with_heat=a.heat, precision=a.precision)
        r1 = torch.zeros(ext.record_bytes(), dtype=torch.uint8, device="cuda")
        lat = []
        for i in range(a.calls + 30):
            if i == 30:
                ext.stage_reset()
            torch.cuda.synchronize()
            t = time.perf_counter()
            ext.extract_batch_device(d1.data_ptr(), 1, r1.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t) * 1e3)
        lat = sorted(lat[30:])
        if mode == "0":
            out["p50_ms"] = round(lat[len(lat) // 2], 4)
            out["p99_ms"] = round(lat[int(len(lat) * 0.99) - 1], 4)
        else:
            out["stage_ms"] = {k: round(v, 4) for k, v in ext.stage_times().items()}
            out["p50_ms_with_stage_events"] = round(lat[len(lat) // 2], 4)
        ext.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
