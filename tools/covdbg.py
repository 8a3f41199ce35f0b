"""Phase timing of the covariance components kernel (SPFE_COV_DEBUG=1), GPU box only."""
import os, sys, numpy as np
os.environ["SPFE_COV_DEBUG"] = "1"; os.environ["SPFE_STAGE_TIMING"] = "1"
sys.path.insert(0, '.')
from sp_orb_slam_amd import weights, synth
from sp_orb_slam_amd.extractor import SPExtractor
H, W = 480, 752
ext = SPExtractor(1000, H, W, weights.synthetic(7, "dense"), with_heat=False)
img = synth.make_image(200, H, W)
for it in range(3):
    ext(img, None)
    d = np.zeros(16, np.uint64); ext._lib.spfe_debug_read(ext._h, b"cov_dbg", 0, d.ctypes.data, 128)
    t = (d[:5].astype(np.int64) - int(d[0])) / 100.0
    print("union %.1f flatten %.1f nxt %.1f replay %.1f us | nd %d maxchain %d maxpops %d workers %d | worker max: walk %.1f moments %.1f stamp %.1f us"
          % (t[1], t[2] - t[1], t[3] - t[2], t[4] - t[3], d[8], d[5], d[6], d[7], d[9] / 100., d[10] / 100., d[11] / 100.),
          "cov stage ms", ext.stage_times()["cov"])
    d[:] = 0
