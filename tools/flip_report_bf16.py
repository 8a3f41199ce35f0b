#!/usr/bin/env python3
"""Flip / margin report of the bf16 mode (BASELINE configs[3]: "bf16 conv path with fp32 NMS"; SURVEY.md §8c: "bf16: ... report
keypoint-set Jaccard").  CPU only, no torch: the oracle's bf16 emulation (oracle_network_bf16: the rounding points of the
bf16 kernels — bf16 weights and activations, f32 accumulation — in plain f32 loops) against the f32 oracle on the same seeded
frames, at the places where a float decides an integer (/root/reference/orb_slam2/src/cv/sp_extractor.cpp):
  * :112  arg-max over the 64 position channels of a cell      -> cells whose arg-max channel differs
  * :122  score >= 0.007                                       -> cells on different sides of the threshold
  * :161-250 nms on the sorted candidates                      -> keypoints present in one result only, keypoint-set Jaccard
and what the back-end does with the descriptors: the L2 distance between the bf16-mode and the f32 descriptor of the SAME
keypoint, set against the matcher's thresholds — 0.3 / 0.7 (sp_matcher.cpp:18-19, TH_LOW / TH_HIGH) and 0.75
(tracker_dust.cpp:122) — i.e. how much of a match budget the precision of the convolutions uses up.

The GPU's bf16 logits differ from the emulation's by the summation order inside a dot product (<= 0.004 of the logit scale,
tests/test_gpu_bf16.py); this report is the statistics of the MODE, the GPU tests hold the kernels to the emulation.

usage: python tools/flip_report_bf16.py [--frames 64] [--out tests/golden/flip_report_bf16.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one_frame(oracle, blob, img, nf):
    H, W = img.shape
    hc, wc = H // 8, W // 8
    semi, coarse, _ = oracle.network(blob, img)
    semi_b, coarse_b = oracle.network_bf16(blob, img)
    am_f, am_b = np.argmax(semi[:, :, :64], -1), np.argmax(semi_b[:, :, :64], -1)
    tf, tb = oracle.tail(semi, H, W), oracle.tail(semi_b, H, W)
    cf, cb = np.zeros(hc * wc, bool), np.zeros(hc * wc, bool)
    cf[tf["cell"]] = True
    cb[tb["cell"]] = True
    rf, rb = oracle.postprocess(semi, coarse, H, W, nf), oracle.postprocess(semi_b, coarse_b, H, W, nf)
    sf = {(int(x), int(y)): i for i, (x, y) in enumerate(rf["kp_xy"])}
    sb = {(int(x), int(y)): i for i, (x, y) in enumerate(rb["kp_xy"])}
    common = [k for k in sb if k in sf]
    a = np.stack([rb["desc"][sb[k]] for k in common]) if common else np.zeros((0, 256), np.float32)
    b = np.stack([rf["desc"][sf[k]] for k in common]) if common else np.zeros((0, 256), np.float32)
    l2 = np.sqrt(((a - b).astype(np.float64) ** 2).sum(1))
    cos = (a.astype(np.float64) * b).sum(1)
    scale = max(1.0, float(np.abs(semi).max()))
    return dict(cells=hc * wc, arg_flips=int((am_f != am_b).sum()), thr_flips=int((cf != cb).sum()),
                kp_only_one_side=len(set(sf) ^ set(sb)), jaccard=len(common) / max(1, len(set(sf) | set(sb))),
                K_f32=len(sf), K_bf16=len(sb), common=len(common), desc_l2=l2, desc_cos=cos,
                desc_max_abs=float(np.abs(a - b).max()) if len(common) else 0.0,
                logit_max_abs_over_scale=float(np.abs(semi - semi_b).max()) / scale,
                coarse_max_abs=float(np.abs(coarse - coarse_b).max()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "flip_report_bf16.json"))
    ap.add_argument("--sizes", default="720x1280,480x752")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    args = ap.parse_args()
    from oracle import oracle
    from sp_orb_slam_amd import synth, weights
    oracle.set_num_threads(args.threads)
    nf = 1000
    t0 = time.time()
    rep = {"what": __doc__.split("\n\n")[0], "num_features": nf, "frames_per_config": args.frames, "seeds": "300 + i",
           "matcher_thresholds": {"TH_LOW sp_matcher.cpp:18": 0.3, "TH_HIGH sp_matcher.cpp:19": 0.7, "patch association tracker_dust.cpp:122": 0.75},
           "configs": {}}
    for size in args.sizes.split(","):
        H, W = (int(v) for v in size.split("x"))
        for det in ("dense", "sparse"):
            blob = weights.synthetic(7, det)
            rows = [one_frame(oracle, blob, synth.make_image(300 + i, H, W), nf) for i in range(args.frames)]
            l2 = np.concatenate([r["desc_l2"] for r in rows])
            cos = np.concatenate([r["desc_cos"] for r in rows])
            jac = np.array([r["jaccard"] for r in rows])
            cells = rows[0]["cells"] * len(rows)
            rep["configs"]["%dx%d_%s" % (W, H, det)] = {
                "frames": len(rows), "cells_per_frame": rows[0]["cells"],
                "arg_flips_total": int(sum(r["arg_flips"] for r in rows)), "arg_flips_per_cell": sum(r["arg_flips"] for r in rows) / cells,
                "thr_flips_total": int(sum(r["thr_flips"] for r in rows)), "thr_flips_per_cell": sum(r["thr_flips"] for r in rows) / cells,
                "keypoints_f32_total": int(sum(r["K_f32"] for r in rows)), "keypoints_bf16_total": int(sum(r["K_bf16"] for r in rows)),
                "keypoints_common_total": int(sum(r["common"] for r in rows)),
                "keypoints_on_one_side_only_total": int(sum(r["kp_only_one_side"] for r in rows)),
                "jaccard_min": float(jac.min()), "jaccard_mean": float(jac.mean()), "jaccard_p05": float(np.percentile(jac, 5)),
                "desc_l2_of_common_keypoints": {"max": float(l2.max()), "mean": float(l2.mean()), "p50": float(np.percentile(l2, 50)),
                                                "p99": float(np.percentile(l2, 99)), "p999": float(np.percentile(l2, 99.9))},
                "desc_l2_share_of_threshold": {"0.3": float(l2.max() / 0.3), "0.7": float(l2.max() / 0.7), "0.75": float(l2.max() / 0.75)},
                "desc_l2_rows_above": {"0.03": int((l2 > 0.03).sum()), "0.1": int((l2 > 0.1).sum()), "0.3": int((l2 > 0.3).sum())},
                "desc_cos_min": float(cos.min()), "desc_max_abs_max": float(max(r["desc_max_abs"] for r in rows)),
                "logit_max_abs_over_scale_max": float(max(r["logit_max_abs_over_scale"] for r in rows)),
                "coarse_max_abs_max": float(max(r["coarse_max_abs"] for r in rows))}
            print(size, det, json.dumps(rep["configs"]["%dx%d_%s" % (W, H, det)])[:400], flush=True)
    rep["seconds"] = round(time.time() - t0, 1)
    with open(args.out, "w") as f:
        json.dump(rep, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
