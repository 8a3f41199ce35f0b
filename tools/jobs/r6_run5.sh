#!/bin/bash
# round 6, job 5: heat maps sent ahead of the record in synchronous host calls (tests, drop-in latency A/B by switch), the cheap load
# hoists in walk / classify / link (single-frame latency A/B of the two libraries)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5; mkdir -p $out
B=tools/microbench/bin
( time timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_selection.py tests/test_gpu_switches.py -x -q ) > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
python - > $out/dropin.txt 2>&1 <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import bench
from sp_orb_slam_amd import synth, weights
blob = weights.synthetic(7, "dense")
fr = synth.make_image(200, 480, 752)
for rep in range(2):
    for flag in ("0", "1"):
        os.environ["SPFE_EARLY_HEAT_COPY"] = flag
        for lazy in (False, True):
            d = bench.dropin_leg(480, 752, 1000, blob, fr, lazy=lazy)
            print("early_heat_copy", flag, "lazy", lazy, d.get("p50"), d.get("p99"), d.get("error"))
PY
cp sp_orb_slam_amd/libspfe.so /tmp/this.so
for rep in 1 2 3; do for which in prev this; do
  [ $which = prev ] && cp $B/libspfe_prev.so sp_orb_slam_amd/libspfe.so || cp /tmp/this.so sp_orb_slam_amd/libspfe.so
  echo -n "$which f32 752x480: " >> $out/latency.txt; python tools/latency_stages.py --calls 600 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['p50_ms'], d['p99_ms'], d['stage_ms'].get('post_side'), d['stage_ms'].get('total'))" >> $out/latency.txt
done; done
cp /tmp/this.so sp_orb_slam_amd/libspfe.so
ONLY=f32 bash tools/microbench/ab_lib.sh $B/libspfe_prev.so 2 > $out/ab_lib.txt 2>&1
tail -3 $out/pytest.log; cat $out/dropin.txt $out/latency.txt $out/ab_lib.txt
