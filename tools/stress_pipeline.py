#!/usr/bin/env python3
"""Stress (GPU box): pipelined device calls (SPFE_FLAG_ASYNC_COV) and the pipelined host path against synchronous calls over
random sizes, batch sizes, feature counts, precisions and detectors — with two side chains in flight forced on (SPFE_TWO_CHAINS=1)
and left to the workload, few generation codes for the covariance maps (SPFE_COV_CAPS field 6: a wrap every few calls), varying
frames per call.  Records must be the same bits.  usage: python tools/stress_pipeline.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from sp_orb_slam_amd import synth, weights  # noqa: E402
from sp_orb_slam_amd.extractor import SPExtractor  # noqa: E402

FIELDS = ("kp_xy", "response", "descriptors", "cov2", "cov2_inv", "occ_grid", "dense_dust", "semi_dust")


def same(a, b, where):
    assert a.status == 0 and a.K == b.K, (where, a.status, a.K, b.K)
    for f in FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), (where, f)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    sizes = [(120, 160), (240, 376), (480, 752), (480, 640), (720, 1280), (136, 200)]
    for case in range(n_cases):
        H, W = sizes[int(rng.integers(len(sizes)))]
        B = int(rng.integers(1, 5)); nf = int(rng.choice([50, 300, 1000])); prec = str(rng.choice(["f32", "bf16"]))
        det = str(rng.choice(["dense", "sparse"])); two = str(rng.choice(["-1", "1", "0"])); gen = int(rng.choice([2, 3, 5, 32766]))
        calls = [int(rng.integers(1, B + 1)) for _ in range(int(rng.integers(5, 9)))]
        picks = [[int(rng.integers(6)) for _ in range(n)] for n in calls]
        seeds = [int(rng.integers(1 << 16)) for _ in range(6)]
        if os.environ.get("STRESS_ONLY") and int(os.environ["STRESS_ONLY"]) != case:
            continue
        print("case %d: %s %dx%d B %d nf %d %s two_chains %s gen_start %d calls %s" % (case, prec, W, H, B, nf, det, two, gen, calls), flush=True)
        blob = weights.synthetic(7, det)
        imgs = [synth.make_image(sd, H, W) for sd in seeds]
        for v in ("SPFE_TWO_CHAINS", "SPFE_COV_CAPS"):
            os.environ.pop(v, None)
        ref_ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, precision=prec)
        ref = [ref_ext.extract_batch([imgs[i] for i in p]) for p in picks]
        ref_ext.close()
        os.environ["SPFE_TWO_CHAINS"] = two
        os.environ["SPFE_COV_CAPS"] = ",,,,,%d" % gen
        # pipelined device calls
        ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, precision=prec, async_cov=True)
        rb = ext.record_bytes()
        d = [torch.from_numpy(im).cuda() for im in imgs]
        stream = torch.cuda.Stream()
        ins = [torch.stack([d[i] for i in p]).contiguous() for p in picks]
        recs = [torch.zeros(len(p) * rb, dtype=torch.uint8, device="cuda") for p in picks]
        torch.cuda.synchronize()   # (the inputs were made on torch's current stream, the calls go to `stream`)
        for p, x, r in zip(picks, ins, recs):
            ext.extract_batch_device(x.data_ptr(), len(p), r.data_ptr(), stream.cuda_stream)
        ext.wait_records(ext.last_ticket(), stream.cuda_stream)
        with torch.cuda.stream(stream):
            host = [r.to("cpu") for r in recs]
        for k, p in enumerate(picks):
            hk = host[k].numpy()
            for i in range(len(p)):
                same(ext.view_record(hk[i * rb:(i + 1) * rb]), ref[k][i], ("device", case, k, i))
        ext.close()
        # pipelined host path
        ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, precision=prec)
        tk, got = [], []
        for p in picks:
            tk.append(ext.submit_batch([imgs[i] for i in p]))
            if len(tk) == 3:
                got.append(ext.collect_batch(tk.pop(0)))
        while tk:
            got.append(ext.collect_batch(tk.pop(0)))
        ext.close()
        for k, p in enumerate(picks):
            for i in range(len(p)):
                same(got[k][i], ref[k][i], ("host", case, k, i))
        print("case %d ok: %s %dx%d B %d nf %d %s two_chains %s gen_start %d calls %s" % (case, prec, W, H, B, nf, det, two, gen, calls), flush=True)
    print("stress_pipeline: %d cases, all records bit-identical" % n_cases)


if __name__ == "__main__":
    main()
