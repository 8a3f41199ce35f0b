import time, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from sp_orb_slam_amd import synth, weights
from sp_orb_slam_amd.extractor import SPExtractor
H, W, nf, B = 480, 752, 1000, 8
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
blob = weights.synthetic(7, "dense")
imgs = [synth.make_image(100 + i, H, W) for i in range(B)]
ext = SPExtractor(nf, H, W, blob, max_batch=B, precision=prec, with_heat=False)
for _ in range(3):
    t = ext.submit_batch(imgs); ext.collect_batch(t, copy=False)
ts, tc = [], []
tick = []
t0 = time.perf_counter()
q = []
for i in range(40):
    a = time.perf_counter(); q.append(ext.submit_batch(imgs)); b = time.perf_counter(); ts.append(b - a)
    if len(q) == 3:
        a = time.perf_counter(); ext.collect_batch(q.pop(0), copy=False); b = time.perf_counter(); tc.append(b - a)
while q:
    ext.collect_batch(q.pop(0), copy=False)
t1 = time.perf_counter()
print(prec, "per call %.3f ms; submit %.3f ms, collect %.3f ms" % ((t1 - t0) / 40 * 1e3, np.median(ts) * 1e3, np.median(tc) * 1e3))
ext.close()

# timeline of a few iterations: when does the host submit / get its batches back
ext = SPExtractor(nf, H, W, blob, max_batch=B, precision=prec, with_heat=False)
for _ in range(3):
    t = ext.submit_batch(imgs); ext.collect_batch(t, copy=False)
q = []; ev = []
t0 = time.perf_counter()
for i in range(8):
    a = time.perf_counter(); q.append(ext.submit_batch(imgs)); b = time.perf_counter()
    ev.append(("submit %d" % i, a - t0, b - t0))
    if len(q) == 3:
        a = time.perf_counter(); tk = q.pop(0); ext.collect_batch(tk, copy=False); b = time.perf_counter()
        ev.append(("collect %d" % (i - 2), a - t0, b - t0))
for name, a, b in ev:
    print("%-10s %7.3f -> %7.3f ms" % (name, a * 1e3, b * 1e3))
ext.close()
