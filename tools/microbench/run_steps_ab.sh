#!/bin/bash
# Does the length of the timed region matter?  (steps, warmup) pairs, interleaved, same box.
Q="--no-cpu-baseline --no-latency --no-stage-table --no-match --no-bf16-leg --no-aten --no-host-path"
for cfg in "--precision bf16 --height 720 --width 1280" "--precision bf16" "--precision f32"; do
  for r in 1 2; do
    for sw in "20 3" "100 10" "300 30"; do
      set -- $sw
      echo -n "steps $1 warmup $2 $cfg: "
      timeout 120 python bench.py $Q $cfg --steps $1 --warmup $2 2>&1 | grep '^{' | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "fps", d["ms_per_step"], "ms  conv1b", d["roofline"]["kernel_ms"], d["roofline"]["frac"])'
    done
  done
done
