"""PCIe-inclusive rate of the host-facing calls (never bench.py's `value`): numpy frames in, host views out."""
import time, numpy as np, sys
sys.path.insert(0, '.')
import torch
from sp_orb_slam_amd import synth, weights
from sp_orb_slam_amd.extractor import SPExtractor
H, W, B = 480, 752, 8
blob = weights.synthetic(7, "dense")
imgs = [synth.make_image(200 + i, H, W) for i in range(B)]
for heat in (False, True):
    ext = SPExtractor(1000, H, W, blob, max_batch=B, with_heat=heat)
    for _ in range(3): ext.extract_batch(imgs)
    t0 = time.perf_counter()
    n = 15
    for _ in range(n): ext.extract_batch(imgs)
    dt = time.perf_counter() - t0
    print("host path (PCIe-inclusive, pageable numpy in, host views out), heat=%s: %.1f frames/s, %.2f ms per 8-frame call" % (heat, n * B / dt, dt / n * 1e3))
    ext.close()
ext = SPExtractor(1000, H, W, blob, max_batch=1, with_heat=False)
for _ in range(5): ext(imgs[0], None)
t0 = time.perf_counter()
for _ in range(100): ext(imgs[0], None)
print("single-frame host call (operator()): %.3f ms" % ((time.perf_counter() - t0) / 100 * 1e3))
