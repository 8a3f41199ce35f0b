#!/bin/bash
# round 6, job 34: conv1a inside conv1b by default in single-frame synchronous f32 calls: tests, latency at three sizes against SPFE_FUSE_CONV1A=0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r34; mkdir -p $out
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
for rep in 1 2 3; do for cfg in "" "--height 480 --width 640" "--height 720 --width 1280"; do for f in 0 auto; do
  [ $f = 0 ] && export SPFE_FUSE_CONV1A=0 || unset SPFE_FUSE_CONV1A
  echo -n "fuse $f [$cfg]: " >> $out/lat.txt; python tools/latency_stages.py --calls 400 $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['p50_ms'], d['p99_ms'])" >> $out/lat.txt
done; done; done
unset SPFE_FUSE_CONV1A
python - <<'PY'
from sp_orb_slam_amd import synth, weights
weights.save("/tmp/w.spfw", weights.synthetic(7, "dense"))
synth.make_image(100, 480, 752).tofile("/tmp/im.raw")
PY
for rep in 1 2 3; do for f in 0 auto; do
  [ $f = 0 ] && export SPFE_FUSE_CONV1A=0 || unset SPFE_FUSE_CONV1A
  echo -n "dropin fuse $f: " >> $out/lat.txt; tools/dropin/bin/dropin_latency /tmp/w.spfw /tmp/im.raw 480 752 1000 400 40 2>&1 | tail -1 >> $out/lat.txt
done; done
tail -4 $out/pytest.log; cat $out/lat.txt
