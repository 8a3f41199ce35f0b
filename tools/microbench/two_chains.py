#!/usr/bin/env python3
"""Would TWO side chains in flight help?  Emulation without touching the library: two handles (each with its own side stream
and buffers) take alternate batches on one launch stream — handle A's chain of batch i then runs beside handle B's
convolutions of batch i + 1 AND A's of batch i + 2, not serialised behind the previous chain.  Prints frames/s of one handle
against two.  usage: two_chains.py <precision> <H> <W> <seed0> [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from sp_orb_slam_amd import parallel, synth, weights
from sp_orb_slam_amd.extractor import SPExtractor

prec, H, W, seed0 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 200
B, nf = 8, 1000
blob = weights.synthetic(7, "dense")
d = torch.from_numpy(synth.make_batch(seed0, B, H, W)).cuda()
stream = torch.cuda.Stream()
def run(nh):
    exts = [SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, async_cov=True, precision=prec) for _ in range(nh)]
    shs = [parallel.ShardedExtractor(e, 1, 0, B) for e in exts]
    for i in range(20):
        shs[i % nh].step(d, stream)
    for s in shs: s.flush(stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        shs[i % nh].step(d, stream)
    for s in shs: s.flush(stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ks = [s.decode(0).K for s in shs]
    for e in exts: e.close()
    return B * steps / dt, ks
for rep in range(2):
    for nh in (1, 2):
        fps, ks = run(nh)
        print("%s %dx%d seed %d: %d handle(s) %.1f frames/s (K %s)" % (prec, W, H, seed0, nh, fps, ks), flush=True)
