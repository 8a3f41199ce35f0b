#!/bin/bash
# round 6, job 47: a single f32 frame read by conv1b straight from the pinned staging buffer (no H2D copy): drop-in and Python call A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r47; mkdir -p $out
python - <<'PY'
from sp_orb_slam_amd import synth, weights
weights.save("/tmp/w.spfw", weights.synthetic(7, "dense"))
synth.make_image(100, 480, 752).tofile("/tmp/im.raw")
PY
for rep in 1 2 3 4; do for z in 0 1; do
  [ $z = 1 ] && export SPFE_ZC=1 || unset SPFE_ZC
  echo -n "zc $z: " >> $out/ab.txt; tools/dropin/bin/dropin_latency /tmp/w.spfw /tmp/im.raw 480 752 1000 400 40 2>&1 | tail -1 >> $out/ab.txt
done; done
unset SPFE_ZC
SPFE_ZC=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 >> $out/ab.txt
cat $out/ab.txt
