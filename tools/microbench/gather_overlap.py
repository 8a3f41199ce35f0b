"""Does the all-gather stream cost the pipelined step anything on ONE GPU?  The same timed loop as bench.py with (a) no
collective, (b) the library's RCCL all-gather with a 1-rank communicator (spfe_allgather_records), (c) a device copy on a
torch stream standing in for the collective.  A waiting stream that shares a hardware queue with the compute stream would
show up as a longer step."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from sp_orb_slam_amd import parallel, synth, weights
from sp_orb_slam_amd.extractor import SPExtractor

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
H, W, nf, B = 480, 752, 1000, 8
blob = weights.synthetic(7, "dense")
frames = [synth.make_image(100 + i, H, W) for i in range(B)]
d_img = torch.from_numpy(np.stack(frames)).cuda()
stream = torch.cuda.Stream()
for mode in ("none", "native", "copy", "none", "native"):
    ext = SPExtractor(nf, H, W, blob, max_batch=B, precision=prec, with_heat=False, async_cov=True)
    if mode == "none":
        sh = parallel.ShardedExtractor(ext, 1, 0, B)
    elif mode == "native":
        sh = parallel.ShardedExtractor(ext, 1, 0, B, native_comm=True)
    else:
        sh = parallel.ShardedExtractor(ext, 1, 0, B, gather_fn=lambda out, loc: out.copy_(loc))
    for _ in range(5):
        sh.step(d_img, stream)
    sh.flush(stream); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        sh.step(d_img, stream)
    sh.flush(stream); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print("%s %-7s %.4f ms per step, %.0f frames/s" % (prec, mode, dt * 1e3, B / dt))
    ext.close()
