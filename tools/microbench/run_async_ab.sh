#!/bin/bash
# Pipelined throughput of the bench workloads (quiet legs) under two environments, interleaved: A="VAR=0" B="VAR=1".
Q="--no-cpu-baseline --no-latency --no-stage-table --no-match --no-bf16-leg --no-aten --no-host-path --steps 100 --warmup 10"
for cfg in "--precision bf16 --height 720 --width 1280" "--precision bf16" "--precision f32"; do
  for r in 1 2 3; do
    for e in "${A:-_A=0}" "${B:-_B=0}"; do
      echo -n "$e $cfg: "
      env $e timeout 120 python bench.py $Q $cfg 2>&1 | grep '^{' | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "fps", d["ms_per_step"], "ms  conv1b", d["roofline"]["kernel_ms"], d["roofline"]["frac"])'
    done
  done
done
