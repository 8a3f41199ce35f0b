#!/usr/bin/env python3
"""Structure of the covariance stage on a workload (GPU box): dirty keypoints, components, chain lengths and
the pops along the longest chain — what bounds cov_replay_kernel (one wavefront per component)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sp_orb_slam_amd import synth, weights  # noqa: E402
from sp_orb_slam_amd.extractor import SPExtractor  # noqa: E402

H, W, nf = (int(a) for a in (sys.argv[1:4] or (480, 752, 1000)))
det = sys.argv[4] if len(sys.argv) > 4 else "dense"
seeds = [int(a) for a in sys.argv[5].split(",")] if len(sys.argv) > 5 else [200, 201, 202]
prec = sys.argv[6] if len(sys.argv) > 6 else "f32"
ext = SPExtractor(nf, H, W, weights.synthetic(7, det), with_heat=False, precision=prec)
for seed in seeds:
    ext(synth.make_image(seed, H, W), None)
    cnt, nxt, workers, npop = (ext.debug_read(n) for n in ("cov_counters", "cov_nxt", "cov_workers", "cov_npop"))
    K = ext.last.K
    chains = []
    for w in workers[:cnt[1]]:
        j, n, pops = int(w), 0, 0
        while j >= 0:
            n += 1
            pops += int(npop[j])
            j = int(nxt[j])
        chains.append((n, pops))
    chains.sort(reverse=True)
    print("seed %d: K %d, dirty %d, components %d, overflow slots %d, pops/keypoint mean %.1f max %d; longest chains (members, pops): %s"
          % (seed, K, cnt[0], cnt[1], cnt[2], npop[:K].mean(), npop[:K].max(), chains[:5]))
ext.close()
