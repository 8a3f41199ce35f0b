#!/bin/bash
# same-box A/B of whole environment SETS over one pipelined workload (default: bf16 1280x720 x 8):
#   tools/microbench/ab_sets.sh [-r reps] "" "SPFE_X=1" "SPFE_X=1 SPFE_Y=2" ...      ("" = the defaults)
# WORKLOAD="--precision bf16" / "" (f32 752x480) / ... overrides the bench arguments; prints value, conv1b frac, ms_per_step
reps=2
if [ "$1" = "-r" ]; then reps=$2; shift 2; fi
cd "$(dirname "$0")/../.."
C="--no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --no-aten --steps ${STEPS:-200} --warmup 20"
W=${WORKLOAD---precision bf16 --height 720 --width 1280}
for rep in $(seq $reps); do
  for s in "$@"; do
    r=$(env $s python bench.py $C $W 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['frac'], d['ms_per_step'])")
    echo "[${s:-defaults}] $r"
  done
done
