#!/usr/bin/env python3
"""Prints the logit error statistics of the bf16 mode against the oracle's bf16 emulation for the
cases of tests/test_gpu_bf16.py (GPU box).  Used to set LOGIT_ATOL / LOGIT_MEAN there."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle  # noqa: E402
from sp_orb_slam_amd import synth, weights  # noqa: E402
from sp_orb_slam_amd.extractor import SPExtractor  # noqa: E402

for H, W, seed, det in [(64, 96, 1, "dense"), (120, 160, 4, "sparse"), (240, 320, 3, "dense"), (480, 752, 100, "dense")]:
    blob = weights.synthetic(7, det)
    img = synth.make_image(seed, H, W)
    ext = SPExtractor(500, H, W, blob, precision="bf16")
    ext(img, None)
    semi, coarse = ext.debug_read("semi"), ext.debug_read("coarse")
    ext.close()
    rsemi, rcoarse = oracle.network_bf16(blob, img)
    fsemi, fcoarse = oracle.network(blob, img)[:2]
    for nm, a, r, f in (("semi", semi, rsemi, fsemi), ("coarse", coarse, rcoarse, fcoarse)):
        d = np.abs(a - r)
        print("%dx%d %-6s %-6s max/scale %.5f  mean/scale %.5f   | bf16 oracle vs f32 oracle: mean/scale %.5f" % (
            W, H, det, nm, d.max() / max(1.0, np.abs(r).max()), d.mean() / max(1.0, np.abs(r).mean()),
            np.abs(r - f).mean() / max(1.0, np.abs(f).mean())))
