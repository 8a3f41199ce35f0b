#!/bin/bash
# round 6, job 39: the record's D2H enqueued by spfe_extract_begin, the early header copy beside the gathered head: tests, drop-in A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r39; mkdir -p $out
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_switches.py tests/test_gpu_stress.py tests/test_abi.py -x -q -m gpu ) > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
python - <<'PY'
from sp_orb_slam_amd import synth, weights
weights.save("/tmp/w.spfw", weights.synthetic(7, "dense"))
synth.make_image(100, 480, 752).tofile("/tmp/im.raw")
PY
cp sp_orb_slam_amd/libspfe.so /tmp/this.so
for rep in 1 2 3 4; do for which in prev this; do
  [ $which = prev ] && cp tools/microbench/bin/libspfe_prev.so sp_orb_slam_amd/libspfe.so || cp /tmp/this.so sp_orb_slam_amd/libspfe.so
  echo -n "$which: " >> $out/dropin.txt; tools/dropin/bin/dropin_latency /tmp/w.spfw /tmp/im.raw 480 752 1000 400 40 2>&1 | tail -1 >> $out/dropin.txt
done; done
cp /tmp/this.so sp_orb_slam_amd/libspfe.so
tail -4 $out/pytest.log; cat $out/dropin.txt
