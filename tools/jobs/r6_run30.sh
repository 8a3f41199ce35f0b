#!/bin/bash
# round 6, job 30: SPFE_SPIN_WAIT (the end of a synchronous host call polled instead of hipStreamSynchronize): drop-in A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r30; mkdir -p $out
python - <<'PY'
from sp_orb_slam_amd import synth, weights
weights.save("/tmp/w.spfw", weights.synthetic(7, "dense"))
synth.make_image(100, 480, 752).tofile("/tmp/im.raw")
PY
for rep in 1 2 3; do for sw in 0 1; do
  echo -n "spin $sw: " >> $out/dropin.txt; SPFE_SPIN_WAIT=$sw tools/dropin/bin/dropin_latency /tmp/w.spfw /tmp/im.raw 480 752 1000 400 40 2>&1 | tail -1 >> $out/dropin.txt
done; done
for sw in 0 1; do echo -n "python host path spin $sw: " >> $out/dropin.txt; SPFE_SPIN_WAIT=$sw python bench.py --no-cpu-baseline --no-bf16-leg --no-match --no-latency --no-stage-table --steps 10 --warmup 3 --verbose-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); hp=d.get('host_path') or {}; print({k:(v.get('p50') if isinstance(v,dict) else v) for k,v in hp.items() if 'single' in k or 'operator' in k})" >> $out/dropin.txt; done
cat $out/dropin.txt
