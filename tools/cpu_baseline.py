#!/usr/bin/env python3
"""CPU baseline table: the C oracle (oracle/spfe_oracle.c — a port; the reference
has no CPU path) on this host's cores, 1 thread and all cores, at the three benchmark resolutions,
split into network / post-processing.  Prints one JSON document; bench.py's `cpu_baseline` is the
all-cores 752x480 row of this table measured in-line.

    python tools/cpu_baseline.py [--frames 5] > profiles/<round>_cpu_baseline.json
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(threads, H, W, frames):
    from oracle import oracle
    from sp_orb_slam_amd import synth, weights

    oracle.set_num_threads(threads)

    blob = weights.synthetic(7, "dense")
    imgs = [synth.make_image(100 + i, H, W) for i in range(frames + 1)]
    oracle.extract(blob, imgs[0], 1000)          # warm-up
    tn = tp = 0.0
    for im in imgs[1:]:
        t0 = time.perf_counter()
        semi, coarse, _ = oracle.network(blob, im)
        t1 = time.perf_counter()
        oracle.postprocess(semi, coarse, H, W, 1000)
        t2 = time.perf_counter()
        tn += t1 - t0
        tp += t2 - t1
    print(json.dumps({"threads": threads, "height": H, "width": W, "frames": frames,
                      "network_ms": round(tn / frames * 1e3, 2), "postprocess_ms": round(tp / frames * 1e3, 2),
                      "fps": round(frames / (tn + tp), 3)}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--worker", nargs=3, type=int, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.worker:
        return worker(args.worker[0], args.worker[1], args.worker[2], args.frames)
    ncpu = os.cpu_count() or 1
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    rows = []
    for (H, W) in ((480, 640), (480, 752), (720, 1280)):
        for thr in sorted({1, min(16, ncpu), min(32, ncpu), min(64, ncpu), min(128, ncpu), ncpu}):
            frames = 2 if thr == 1 else args.frames
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--frames", str(frames), "--worker",
                                  str(thr), str(H), str(W)], capture_output=True, text=True, check=True)
            rows.append(json.loads(out.stdout.strip().splitlines()[-1]))
    print(json.dumps({"what": "oracle/spfe_oracle.c (C restatement, OpenMP), dense synthetic weights, 1000 features",
                      "kind": "port", "cpu": model, "logical_cpus": ncpu, "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
