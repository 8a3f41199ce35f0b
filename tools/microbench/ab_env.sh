#!/bin/bash
# same-box A/B of one environment knob over the three pipelined workloads: usage tools/microbench/ab_env.sh VAR "v1 v2" [reps]
var=$1; vals=$2; reps=${3:-2}
cd "$(dirname "$0")/../.."
C="--no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --steps 200 --warmup 20"
for rep in $(seq $reps); do
  for v in $vals; do
    for cfg in "f32:" "bf16_720p:--precision bf16 --height 720 --width 1280" "bf16_752:--precision bf16"; do
      n=${cfg%%:*}; a=${cfg#*:}
      [ -n "$ONLY" ] && [ "$ONLY" != "$n" ] && continue
      r=$(env $var=$v python bench.py $C $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['frac'])")
      echo "$var=$v $n $r"
    done
  done
done
