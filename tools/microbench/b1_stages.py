import os, sys, json
os.environ["SPFE_STAGE_TIMING"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from sp_orb_slam_amd.extractor import SPExtractor
from sp_orb_slam_amd import weights, synth
for prec, H, W in (("f32", 480, 752), ("bf16", 720, 1280)):
    blob = weights.synthetic(7, "dense")
    ext = SPExtractor(1000, H, W, blob, max_batch=1, with_heat=False, precision=prec)
    d = torch.from_numpy(synth.make_batch(300, 1, H, W)).cuda()
    rec = torch.empty(ext.record_bytes(), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    for _ in range(20):
        ext.extract_batch_device(d.data_ptr(), 1, rec.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    ext.stage_reset()
    for _ in range(50):
        ext.extract_batch_device(d.data_ptr(), 1, rec.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    print(prec, H, W, json.dumps({k: round(v, 4) for k, v in ext.stage_times().items()}))
    ext.close()
