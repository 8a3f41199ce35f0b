"""Single-solve latency of spfe_align_dust_record_device on the scene of tests/golden/dust_std0.npz (160 map points,
752x480): the call a tracker makes once per frame.  `python tools/microbench/dust_time.py [reps]`.
With a libspfe.so whose dust.hip was compiled with -DSPFE_DUST_PROBE (tools/microbench/build_probes.sh) the kernel also
prints the cycle counts of its phases."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from sp_orb_slam_amd import parallel, weights  # noqa: E402
from sp_orb_slam_amd.extractor import DUST_OUT_BYTES, SPExtractor  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
H, W, nf = 480, 752, 1000
gz = np.load(os.path.join(ROOT, "tests", "golden", "dust_std0.npz"))
fx, fy, cx, cy = (np.float32(v) for v in gz["intr"])
ext = SPExtractor(nf, H, W, weights.synthetic(7, "dense"), max_batch=1, with_heat=False)
lay = parallel.RecordLayout(H, W, nf)
d_rec = torch.zeros(ext.record_bytes(), dtype=torch.uint8, device="cuda")
d_rec[lay.off_dd:lay.off_dd + gz["dust"].size * 4] = torch.from_numpy(gz["dust"].reshape(-1).view(np.uint8).copy()).cuda()
n = len(gz["pts"])
d_pts, d_T = torch.from_numpy(gz["pts"]).cuda(), torch.from_numpy(gz["Tcw_init"].reshape(16)).cuda()
d_out = torch.zeros(DUST_OUT_BYTES, dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()


def once():
    ext.align_dust_record_device(d_rec.data_ptr(), d_pts.data_ptr(), n, d_T.data_ptr(), d_out.data_ptr(), fx, fy, cx, cy,
                                 stream=st.cuda_stream)


for _ in range(3):
    once()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(reps):
    once()
e1.record(st)
torch.cuda.synchronize()
g = ext.decode_dust_out(d_out.cpu().numpy(), n)
t0 = time.perf_counter()
r = oracle.align_dust(gz["dust"], gz["pts"], gz["Tcw_init"], fx, fy, cx, cy)
t_cpu = time.perf_counter() - t0
print("dust single solve: %.1f us (GPU, %d points, %d iterations) | CPU oracle %.1f us | pose max-abs vs fixture %.3g, vs oracle %.3g | "
      "iterations equal %s, inlier flags equal %s" %
      (e0.elapsed_time(e1) / reps * 1e3, n, g["iterations"], t_cpu * 1e6,
       np.abs(g["Tcw"].astype(np.float64) - gz["pose64"]).max(), np.abs(g["Tcw"] - r["Tcw"]).max(),
       g["iterations"] == int(gz["iterations"]), np.array_equal(g["inlier"], gz["inlier"])))
