#!/usr/bin/env python3
"""Generate golden fixtures for the SuperPoint extraction path.

Run ONLY in the build container (needs torch; nothing here travels except the
.npz files it writes).  It is an INDEPENDENT second restatement of the
reference path used to pin the C oracle:

  * the network, detector tail and descriptor sampling issue the SAME ATen op
    sequence as SPFrontend::forward (/root/reference/orb_slam2/src/cv/
    sp_extractor.cpp:79-159) on PyTorch-CPU: conv2d/relu/max_pool2d/softmax/
    max/gather/masked_select/clamp/log/pixel_shuffle/grid_sampler_2d/norm;
  * the host glue (to_heat :461-474, sort :489-498, nms :161-250,
    computeCovariance :252-340) is ported line by line to Python/numpy.

The reference itself cannot run here (CUDA hard-wired :73,:134,:348-351, weights
missing), so these fixtures are NOT outputs of the reference: parity stays
"unpinned" in the sense of SURVEY.md §8c; they pin the oracle's conventions
(layouts, channel<->pixel mapping, tie rules, sampling coordinates) against
ATen's implementation of each op.

Usage: python tests/golden/make_golden.py [case ...]   (writes tests/golden/*.npz)
"""
import os
import sys
from collections import deque

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from sp_orb_slam_amd import synth, weights  # noqa: E402

torch.set_num_threads(8)
torch.backends.mkldnn.enabled = True


from tools.aten_path import forward  # noqa: E402  (the ATen op sequence; shared with bench.py)


def to_heat(heat_log):
    """:461-474 (affine MatExpr evaluation, float mul then float add)."""
    img = -heat_log
    mn, mx = float(img.min()), float(img.max())
    inv = 1.0 / (mx - mn)
    a_h, b_h = np.float32(-inv), np.float32(-mn * inv)
    a_i, b_i = np.float32(inv), np.float32(mx * inv)
    heat = (heat_log * a_h).astype(np.float32) + b_h
    heat_inv = (heat_log * a_i).astype(np.float32) + b_i
    return heat.astype(np.float32), heat_inv.astype(np.float32)


def sort_desc(score):
    """:489-498 with the build's tie rule: descending score, ascending index."""
    return np.lexsort((np.arange(len(score)), -score.astype(np.float64))).astype(np.int32)


def nms(pts_sorted, num_features, W, H, border=8, dist=4):
    """:161-250, literal."""
    n = len(pts_sorted)
    grid = np.zeros((H + 2 * dist, W + 2 * dist), np.uint8)
    inds = np.zeros((H, W), np.uint16)
    occ = np.full((H // 8, W // 8), -1, np.int16)
    pr = [(int(p[0]), int(p[1])) for p in pts_sorted]
    for i, (u, v) in enumerate(pr):
        grid[v + dist, u + dist] = 1
        inds[v, u] = i
    n_feature = 0
    for (u, v) in pr:
        uu, vv = u + dist, v + dist
        if grid[vv, uu] != 1:
            continue
        grid[vv - dist:vv + dist + 1, uu - dist:uu + dist + 1] = 0
        grid[vv, uu] = 2
        n_feature += 1
        if n_feature > num_features:
            break
    kps, sel = [], []
    for v in range(H + dist):
        for u in range(W + dist):
            if u - dist >= W - border or u - dist < border or v - dist >= H - border or v - dist < border:
                continue
            if grid[v, u] == 2:
                occ[(v - dist) // 8, (u - dist) // 8] = len(kps)
                s = int(inds[v - dist, u - dist])
                kps.append(pr[s])
                sel.append(s)
    return np.array(kps, np.float32).reshape(-1, 2), np.array(sel, np.int32), occ


def compute_covariance(heat, kps):
    """:252-340, literal (float32 arithmetic like Eigen::Vector2f)."""
    H, W = heat.shape
    fresh = np.ones((H, W), np.uint8)
    cov, cinv, resp = [], [], []
    f32 = np.float32
    for (x, y) in kps:
        uu, vv = int(x), int(y)
        resp.append(heat[vv, uu])
        q = deque([(uu, vv)])
        d2, sc = [], []
        while q:
            u, v = q.popleft()
            fresh[v, u] = 0
            d2.append((f32(u - uu) * f32(u - uu), f32(v - vv) * f32(v - vv)))
            c = heat[v, u]
            sc.append(c)
            for (u_, v_, ok) in ((u - 1, v, u - 1 > 0), (u, v - 1, v - 1 > 0),
                                 (u + 1, v, u + 1 < W), (u, v + 1, v + 1 < H)):
                if ok and fresh[v_, u_] and heat[v_, u_] > 0.0 and heat[v_, u_] < c:
                    q.append((u_, v_))
        s = f32(0)
        for v_ in sc:
            s = f32(s + v_)
        cx, cy = f32(0), f32(0)
        for (dx, dy), v_ in zip(d2, sc):
            wgt = f32(v_ / s)
            cx = f32(cx + f32(wgt * dx))
            cy = f32(cy + f32(wgt * dy))
        cx, cy = max(cx, f32(1)), max(cy, f32(1))
        cov.append((cx, cy))
        cinv.append((f32(1) / cx, f32(1) / cy))
    return (np.array(cov, np.float32).reshape(-1, 2), np.array(cinv, np.float32).reshape(-1, 2),
            np.array(resp, np.float32))


def extract(named, img, num_features, cuda_scalar_div=False):
    """SPExtractor::operator() (:361-514).  cuda_scalar_div: tools/aten_path.forward's CUDA form of `pixels.div(W / 2.0)`."""
    H, W = img.shape
    f = forward(named, img, cuda_scalar_div)
    pts = f["pixels_in"].T.copy()  # [N,2]  (:458)
    heat, heat_inv = to_heat(f["heat_log"])
    order = sort_desc(f["score"])
    pts_sorted = pts[order]
    desc_sorted = f["desc"][order]
    kps, sel, occ = nms(pts_sorted, num_features, W, H)
    desc = desc_sorted[sel] if len(sel) else np.zeros((0, 256), np.float32)
    cov, cinv, resp = compute_covariance(heat_inv, kps)
    f.update(heat=heat, heat_inv=heat_inv, order=order, kp_xy=kps, kp_src=sel, occ_grid=occ,
             kp_desc=desc, cov2=cov, cov2_inv=cinv, response=resp)
    return f


CASES = [
    # name, H, W, image seed, weight seed, detector, num_features, full?
    ("g64x96_dense", 64, 96, 1, 7, "dense", 20, True),
    ("g64x96_sparse", 64, 96, 2, 7, "sparse", 100, True),
    ("g128x160_sparse", 128, 160, 5, 11, "sparse", 200, True),
    ("g480x752_dense", 480, 752, 100, 7, "dense", 1000, False),
    ("g480x640_sparse", 480, 640, 1, 7, "sparse", 1000, False),
    ("g720x1280_sparse", 720, 1280, 300, 7, "sparse", 1000, False),   # BASELINE configs[3] size
    # `pixels.div(W / 2.0)` (:137-138) as libtorch-1.6 CUDA evaluates it — a * (1.0f / b) — the device the reference
    # hard-wires (:73): the form the oracle and the kernels pin.  W / 2 = 188 and 376: the reciprocal is
    # inexact, so the two forms differ in the last bit of about a third of the sampling coordinates
    ("g64x376_dense_cudadiv", 64, 376, 6, 7, "dense", 150, True),
    ("g480x752_dense_cudadiv", 480, 752, 100, 7, "dense", 1000, False),
]


def main():
    only = set(sys.argv[1:])   # optional: names of the cases to (re)generate
    for name, H, W, iseed, wseed, det, nf, full in CASES:
        if only and name not in only:
            continue
        img = synth.make_image(iseed, H, W)
        blob = weights.synthetic(wseed, det)
        cuda_div = name.endswith("_cudadiv")
        out = extract(weights.to_named_tensors(blob), img, nf, cuda_div)
        meta = dict(H=H, W=W, image_seed=iseed, weight_seed=wseed, detector=det, num_features=nf, cuda_scalar_div=int(cuda_div))
        keep = dict(image=img, kp_xy=out["kp_xy"].astype(np.int16), occ_grid=out["occ_grid"],
                    n_candidates=np.int32(len(out["score"])), response=out["response"],
                    cov2=out["cov2"], cov2_inv=out["cov2_inv"],
                    dense_dust=out["dense_dust"], semi_dust=out["semi_dust"], **{
                        "meta_" + k: np.array(v) for k, v in meta.items()})
        if cuda_div:
            # how much the fixture discriminates: the same frame through ATen-CPU's true division
            cpu = forward(weights.to_named_tensors(blob), img, False)
            keep["meta_desc_max_abs_diff_to_cpu_div_form"] = np.float32(np.abs(cpu["desc"] - out["desc"]).max())
            keep["meta_desc_rows_differing_from_cpu_div_form"] = np.int32((np.abs(cpu["desc"] - out["desc"]).max(1) > 1e-7).sum())
        if full:
            keep.update(semi=out["semi"], coarse_raw=out["coarse_raw"], heat_log=out["heat_log"],
                        heat=out["heat"], heat_inv=out["heat_inv"], cand_xy=out["pixels_in"].T.copy(),
                        cand_score=out["score"], cand_desc=out["desc"], order=out["order"],
                        kp_src=out["kp_src"], kp_desc=out["kp_desc"])
        else:
            # large case: keep it small — every 16th descriptor row, a heat row, score stats
            keep.update(kp_desc_sub=out["kp_desc"][::16].copy(), heat_row=out["heat"][H // 2].copy(),
                        cand_score_sorted_head=np.sort(out["score"])[::-1][:64].copy(),
                        semi_sub=out["semi"][::7, ::9].copy(), coarse_sub=out["coarse_raw"][::13, ::11].copy())
            del keep["image"]  # regenerated from the seed (synth is integer-exact)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **keep)
        print("%-18s K=%4d N=%5d  %7.1f kB" % (name, len(out["kp_xy"]), len(out["score"]),
                                             os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
