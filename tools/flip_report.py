#!/usr/bin/env python3
"""Flip / margin report ("end-to-end fp32: report #cells with flipped argmax/threshold (expected ~0 for margin
> 1e-5)").  BUILD CONTAINER ONLY (imports torch for the ATen-CPU op sequence; nothing here travels but the JSON it writes).

For every seeded frame, two independent f32 evaluations of the same network — the CPU oracle (oracle/spfe_oracle.c: the
contract's k-ordered fmaf chains = the GPU kernels' bits) and the ATen-CPU op sequence of SPFrontend::forward
(tools/aten_path.py: MKL-DNN's summation order) — are compared at the three places where a float decides an integer
(/root/reference/orb_slam2/src/cv/sp_extractor.cpp):
  * :112  arg-max over the 64 position channels of a cell      -> cells whose arg-max channel differs
  * :122  score >= 0.007                                       -> cells on different sides of the threshold
  * :161-250 nms on the sorted candidates                      -> keypoints present in one result only
and the margins that say how close the frame was to a flip: the smallest gap between the two largest position logits of a
cell, and the smallest |score - 0.007|.  The logits of the two evaluations differ by ~1e-5 (different summation order over
K = 576 ... 1152 products), so a cell can only flip when its margin is of that size.

usage: python tools/flip_report.py [--frames 64] [--out tests/golden/flip_report.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def one_frame(named, blob, img, nf, aten_path, oracle, mg):
    H, W = img.shape
    hc, wc = H // 8, W // 8
    a = aten_path.forward(named, img, True)
    semi, coarse, _ = oracle.network(blob, img)
    # arg-max (:112): lowest index on ties, both sides
    am_o = np.argmax(semi[:, :, :64], -1).astype(np.int32)
    am_a = a["argmax_map"]
    arg_flips = int((am_o != am_a).sum())
    # threshold (:122)
    t = oracle.tail(semi, H, W)
    cand_o = np.zeros(hc * wc, bool)
    cand_o[t["cell"]] = True
    cand_a = (a["score_map"] >= np.float32(0.007)).reshape(-1)
    thr_flips = int((cand_o != cand_a).sum())
    # margins from the oracle's logits / scores
    part = np.partition(semi[:, :, :64], 62, axis=-1)
    top2_gap = (part[:, :, 63] - part[:, :, 62]).reshape(-1)
    e = np.exp((semi - semi.max(-1, keepdims=True)).astype(np.float64))
    score = (e[:, :, :64].max(-1) / e.sum(-1)).reshape(-1)
    thr_gap = np.abs(score - 0.007)
    # nms outcome (:161-250): the oracle's keypoints against the literal Python nms on ATen's candidates
    out = oracle.postprocess(semi, coarse, H, W, nf)
    order = mg.sort_desc(a["score"])
    kps_a, _, _ = mg.nms(a["pixels_in"].T[order], nf, W, H)
    so = {(int(x), int(y)) for x, y in out["kp_xy"]}
    sa = {(int(x), int(y)) for x, y in kps_a}
    logit_diff = float(np.abs(semi - a["semi"]).max())
    return dict(cells=hc * wc, arg_flips=arg_flips, thr_flips=thr_flips, kp_only_one_side=len(so ^ sa), K=len(so),
                n_candidates=int(cand_o.sum()), top2_gap_min=float(top2_gap.min()), thr_gap_min=float(thr_gap.min()),
                top2_below={k: int((top2_gap < float(k)).sum()) for k in ("1e-6", "1e-5", "1e-4", "1e-3")},
                thr_below={k: int((thr_gap < float(k)).sum()) for k in ("1e-7", "1e-6", "1e-5", "1e-4")},
                logit_max_abs_diff=logit_diff,
                # the flipped cells' own margins (a flip needs a margin of the size of the logit difference)
                arg_flip_gaps=[float(v) for v in np.sort(top2_gap[(am_o != am_a).reshape(-1)])[:8]],
                thr_flip_gaps=[float(v) for v in np.sort(thr_gap[cand_o != cand_a])[:8]])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "flip_report.json"))
    ap.add_argument("--sizes", default="480x640,480x752,720x1280")
    a = ap.parse_args()
    import torch
    torch.set_num_threads(os.cpu_count() or 8)
    import make_golden as mg
    from oracle import oracle
    from sp_orb_slam_amd import synth, weights
    from tools import aten_path
    nf = 1000
    rep = {"what": __doc__.split("\n\n")[0], "num_features": nf, "frames_per_config": a.frames, "seeds": "1000 + i",
           "torch": torch.__version__, "configs": {}}
    t0 = time.time()
    for size in a.sizes.split(","):
        H, W = (int(v) for v in size.split("x"))
        for det in ("dense", "sparse"):
            blob = weights.synthetic(7, det)
            named = weights.to_named_tensors(blob)
            rows = []
            for i in range(a.frames):
                rows.append(one_frame(named, blob, synth.make_image(1000 + i, H, W), nf, aten_path, oracle, mg))
            key = "%dx%d_%s" % (W, H, det)
            agg = dict(frames=len(rows), cells_per_frame=rows[0]["cells"],
                       arg_flips_total=sum(r["arg_flips"] for r in rows), thr_flips_total=sum(r["thr_flips"] for r in rows),
                       frames_with_any_flip=sum(1 for r in rows if r["arg_flips"] or r["thr_flips"]),
                       keypoints_on_one_side_only_total=sum(r["kp_only_one_side"] for r in rows),
                       frames_with_different_keypoints=sum(1 for r in rows if r["kp_only_one_side"]),
                       keypoints_total=sum(r["K"] for r in rows), candidates_total=sum(r["n_candidates"] for r in rows),
                       top2_gap_min=min(r["top2_gap_min"] for r in rows), thr_gap_min=min(r["thr_gap_min"] for r in rows),
                       cells_with_top2_gap_below={k: sum(r["top2_below"][k] for r in rows) for k in rows[0]["top2_below"]},
                       cells_with_threshold_gap_below={k: sum(r["thr_below"][k] for r in rows) for k in rows[0]["thr_below"]},
                       logit_max_abs_diff=max(r["logit_max_abs_diff"] for r in rows),
                       flipped_cells_top2_gaps=sorted(v for r in rows for v in r["arg_flip_gaps"])[:16],
                       flipped_cells_threshold_gaps=sorted(v for r in rows for v in r["thr_flip_gaps"])[:16])
            rep["configs"][key] = agg
            print("%-18s frames %d  arg flips %d  thr flips %d  kp diff %d  logit diff %.2e  top2 min %.2e  thr min %.2e  (%.0f s)" % (
                key, len(rows), agg["arg_flips_total"], agg["thr_flips_total"], agg["keypoints_on_one_side_only_total"],
                agg["logit_max_abs_diff"], agg["top2_gap_min"], agg["thr_gap_min"], time.time() - t0), flush=True)
            json.dump(rep, open(a.out, "w"), indent=1)
    rep["seconds"] = round(time.time() - t0, 1)
    json.dump(rep, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
