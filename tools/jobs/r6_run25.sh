#!/bin/bash
# round 6, job 25: spfe_extract_rows (descriptor rows copied beside the covariance) on top of spfe_set_map_buffers (the drop-in's heat_ / heat_inv_ storage as the D2H destination): tests, and the
# drop-in's operator() in place / copied (three-part call) / lazy against the binary of before both
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r25; mkdir -p $out
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
  for mode in "" lazy; do
    echo -n "old [$mode]: " >> $out/dropin.txt; $B/dropin_latency_old /tmp/w.spfw /tmp/im.raw 480 752 1000 400 40 $mode 2>&1 | tail -1 >> $out/dropin.txt
  done
  for mode in "" lazy copy; do
    echo -n "new [$mode]: " >> $out/dropin.txt; tools/dropin/bin/dropin_latency /tmp/w.spfw /tmp/im.raw 480 752 1000 400 40 $mode 2>&1 | tail -1 >> $out/dropin.txt
  done
done
python tools/latency_stages.py --calls 400 2>/dev/null | head -c 600 >> $out/dropin.txt
tail -5 $out/pytest.log; cat $out/dropin.txt
