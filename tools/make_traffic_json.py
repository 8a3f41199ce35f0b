#!/usr/bin/env python3
"""profiles/<name>_traffic.json from a PMC summary (tools/profile_round.sh): HBM bytes per launch of the dominant
kernel = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 — FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950
(128-byte requests tallied at 64 B) — stamped with the sha256 of the kernel sources it was measured on, so that
bench.py nulls `roofline.traffic` when the kernel has changed since.

usage: tools/make_traffic_json.py <pmc_summary.txt> <kernel substring> <H> <W> <B> <source.hip>... > profiles/x.json"""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
summ, kern, H, W, B = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
srcs = [a for a in sys.argv[6:] if not a.startswith("share=")]
# share=<x>: the measured launch covers that part of conv1b's output rows (the f32 list is cut in a 16-row and an 8-row launch)
share = float(([a[6:] for a in sys.argv[6:] if a.startswith("share=")] or ["1"])[0])
vals = {}
for line in open(summ):
    if kern in line:
        m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+([\d.]+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)", line)
        if m:   # the template may be shared by several layers: the largest dispatch is conv1b
            vals[m.group(1)] = float(m.group(5))
fetch_kb, write_kb = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
el = 4 if "f32" in kern else 2
if "true,2" in kern:   # conv1a fused into the producers: the kernel reads the u8 image, not conv1a's activation
    alg = B * (H * W + (H // 2) * (W // 2) * 64 * el) + 64 * 576 * el + 64 * 16 * 2 + 64 * 4
else:
    alg = share * B * (H * W * 64 * el + (H // 2) * (W // 2) * 64 * el) + 64 * 576 * el
out = {
    "kernel": "%s (conv1b), %d frames %dx%d per launch" % (kern, B, W, H),
    "workload_hwb": [H, W, B], "share_of_conv1b_in_this_launch": share,
    "fetch_size_kb_max_dispatch": fetch_kb, "write_size_kb_max_dispatch": write_kb,
    "fetch_correction": "x2 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section)",
    "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024),
    "algorithmic_bytes_per_launch": int(alg),
    "source": os.path.relpath(summ, ROOT),
    "kernel_source_sha16": {s: hashlib.sha256(open(os.path.join(ROOT, "sp_orb_slam_amd", "csrc", s), "rb").read()).hexdigest()[:16] for s in srcs},
    "measured_at_commit": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
}
json.dump(out, sys.stdout, indent=1)
print()
