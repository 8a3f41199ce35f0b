#!/bin/bash
# round 6, job 38: one drop-in call from both sides (HIP API calls, copies, kernels): where the host's 0.13 ms above the device's span go
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r38; mkdir -p $out
python - <<'PY'
from sp_orb_slam_amd import synth, weights
weights.save("/tmp/w.spfw", weights.synthetic(7, "dense"))
synth.make_image(100, 480, 752).tofile("/tmp/im.raw")
PY
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace -d $out/tr -o trace -- tools/dropin/bin/dropin_latency /tmp/w.spfw /tmp/im.raw 480 752 1000 30 10 > $out/tr.log 2>&1
python tools/rocpd_host_timeline.py $out/tr/*.db -3 > $out/host_timeline.txt 2>&1
rm -rf $out/tr
tail -2 $out/tr.log | cut -c1-200; cat $out/host_timeline.txt | cut -c1-140
