#!/usr/bin/env python3
"""Stress: the gathered descriptor branch against the dense one over random sizes, batch sizes, feature counts, detectors and
call patterns (GPU box).  Records must be the same bits.  usage: python tools/stress_sparse.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sp_orb_slam_amd import synth, weights  # noqa: E402
from sp_orb_slam_amd.extractor import SPExtractor  # noqa: E402


def run(flag, prec, H, W, B, nf, det, sets, pattern):
    os.environ["SPFE_SPARSE_DB"] = flag
    ext = SPExtractor(nf, H, W, weights.synthetic(7, det), max_batch=B, with_heat=False, precision=prec)
    out = []
    if pattern == "sync":
        for s in sets:
            out += ext.extract_batch(s)
    else:
        tk = []
        for s in sets:
            tk.append(ext.submit_batch(s))
            if len(tk) == 3:
                out += ext.collect_batch(tk.pop(0))
        while tk:
            out += ext.collect_batch(tk.pop(0))
    ext.close()
    return out


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for case in range(n_cases):
        prec = "f32" if rng.random() < 0.6 else "bf16"
        H = int(rng.integers(8, 70)) * 8
        W = int(rng.integers(8, 100)) * 8
        B = int(rng.integers(1, 5))
        nf = int(rng.choice([20, 150, 500, 1000, 3000]))
        det = "dense" if rng.random() < 0.7 else "sparse"
        pattern = "sync" if rng.random() < 0.5 else "pipe"
        ncalls = int(rng.integers(2, 6))
        sets = [[synth.make_image(int(rng.integers(0, 10000)), H, W) for _ in range(int(rng.integers(1, B + 1)))] for _ in range(ncalls)]
        a = run("0", prec, H, W, B, nf, det, sets, pattern)
        b = run("1", prec, H, W, B, nf, det, sets, pattern)
        ok = len(a) == len(b) and all(
            x.K == y.K and np.array_equal(x.kp_xy, y.kp_xy) and
            np.array_equal(x.descriptors.view(np.uint32), y.descriptors.view(np.uint32)) and
            np.array_equal(x.cov2.view(np.uint32), y.cov2.view(np.uint32)) for x, y in zip(a, b))
        bad += not ok
        print("case %3d %s %4dx%-4d B=%d nf=%-4d %-6s %-4s calls=%d frames=%d K0=%d : %s" %
              (case, prec, W, H, B, nf, det, pattern, ncalls, len(a), a[0].K, "ok" if ok else "MISMATCH"), flush=True)
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
