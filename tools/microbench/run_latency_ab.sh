#!/bin/bash
# Batch-1 latency leg of bench.py under two environments, interleaved: A="VAR=0" B="VAR=1".
Q="--no-cpu-baseline --no-stage-table --no-match --no-bf16-leg --no-aten --no-host-path --steps 50 --warmup 10"
for cfg in "--precision f32" "--precision bf16 --height 720 --width 1280" "--precision f32 --detector sparse"; do
  for r in 1 2; do
    for e in "${A:-_A=0}" "${B:-_B=0}"; do
      echo -n "$e $cfg: "
      env $e timeout 160 python bench.py $Q $cfg 2>&1 | grep '^{' | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("fps", d["value"], "latency", d["latency_batch1_ms"])'
    done
  done
done
