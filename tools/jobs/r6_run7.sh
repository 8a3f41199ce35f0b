#!/bin/bash
# round 6, job 7: where conv1b's (fused conv1a) non-MFMA cycles go (timing / ablation builds of the probe), and the early descriptor-row D2H
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r7; mkdir -p $out
B=tools/microbench/bin
for v in w_t w_t_a2 w_t_a4 w_t_a5; do
  for a in "720 1280 8 64 1 100 fuse" "360 640 8 64 0 100"; do echo "== $v $a" >> $out/probe.txt; PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v $a 2>&1 | grep -v "^(fuse" >> $out/probe.txt; done
done
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q ) > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
python - > $out/dropin.txt 2>&1 <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import bench
from sp_orb_slam_amd import synth, weights
blob = weights.synthetic(7, "dense")
fr = synth.make_image(200, 480, 752)
for rep in range(3):
    for flag in ("0", "1"):
        os.environ["SPFE_EARLY_HEAT_COPY"] = flag
        for lazy in (False, True):
            d = bench.dropin_leg(480, 752, 1000, blob, fr, lazy=lazy)
            print("early_copies", flag, "lazy", lazy, d.get("p50"), d.get("p99"), d.get("error"))
PY
python - > $out/hostcall.txt 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from sp_orb_slam_amd import synth, weights
from sp_orb_slam_amd.extractor import SPExtractor
blob = weights.synthetic(7, "dense")
fr = synth.make_image(200, 480, 752)
for rep in range(3):
    for flag in ("0", "1"):
        os.environ["SPFE_EARLY_HEAT_COPY"] = flag
        for heat in (False, True):
            ext = SPExtractor(1000, 480, 752, blob, with_heat=heat)
            for _ in range(30): ext(fr, None)
            ts = []
            for _ in range(300):
                t0 = time.perf_counter(); ext(fr, None); ts.append(time.perf_counter() - t0)
            ts.sort(); ext.close()
            print("early_copies", flag, "heat", heat, "python operator() p50 %.4f ms" % (ts[150] * 1e3))
PY
cat $out/probe.txt; tail -3 $out/pytest.log; cat $out/dropin.txt $out/hostcall.txt
