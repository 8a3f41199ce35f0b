#!/bin/bash
# round 6, job 23 (= 22 again, heat and heat_inv waited for separately): the synchronous call in three parts (spfe_extract_begin / _maps / _finish): tests, and the drop-in's
# operator() with its map copies beside the device's chain against the binary built before the change
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r23; mkdir -p $out
B=tools/microbench/bin
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_switches.py tests/test_abi.py -x -q -m gpu ) > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
python - <<'PY'
import numpy as np
from sp_orb_slam_amd import synth, weights
weights.save("/tmp/w.spfw", weights.synthetic(7, "dense"))
synth.make_image(100, 480, 752).tofile("/tmp/im.raw")
PY
for rep in 1 2 3; do
  for which in old new; do
    [ $which = old ] && exe=$B/dropin_latency_old || exe=tools/dropin/bin/dropin_latency
    for mode in "" lazy; do
      echo -n "$which [$mode]: " >> $out/dropin.txt; $exe /tmp/w.spfw /tmp/im.raw 480 752 1000 400 40 $mode 2>&1 | tail -1 >> $out/dropin.txt
    done
  done
done
tail -5 $out/pytest.log; cat $out/dropin.txt
