"""Per-stage times (SPFE_STAGE_TIMING=1) of small synchronous calls: python tools/microbench/b1_stages.py [prec H W B ...]"""
import os, sys, json
os.environ["SPFE_STAGE_TIMING"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from sp_orb_slam_amd.extractor import SPExtractor
from sp_orb_slam_amd import weights, synth
CASES = [("f32", 480, 752, 1), ("bf16", 720, 1280, 1)]
if len(sys.argv) > 1:   # prec H W B ...
    a = sys.argv[1:]
    CASES = [(a[i], int(a[i + 1]), int(a[i + 2]), int(a[i + 3])) for i in range(0, len(a), 4)]
for prec, H, W, B in CASES:
    blob = weights.synthetic(7, "dense")
    ext = SPExtractor(1000, H, W, blob, max_batch=B, with_heat=False, precision=prec)
    d = torch.from_numpy(synth.make_batch(300, B, H, W)).cuda()
    rec = torch.empty(B * ext.record_bytes(), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    for _ in range(20):
        ext.extract_batch_device(d.data_ptr(), B, rec.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    ext.stage_reset()
    for _ in range(50):
        ext.extract_batch_device(d.data_ptr(), B, rec.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    print(prec, H, W, B, json.dumps({k: round(v, 4) for k, v in ext.stage_times().items()}))
    ext.close()
