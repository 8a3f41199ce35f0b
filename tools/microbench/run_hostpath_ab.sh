#!/bin/bash
# Host-path leg (pageable frames in -> host views out) under two environments, interleaved: A="VAR=0" B="VAR=1".
Q="--no-cpu-baseline --no-latency --no-stage-table --no-match --no-bf16-leg --no-aten --steps 200 --warmup 20"
for cfg in "--precision bf16 --height 720 --width 1280" "--precision bf16" "--precision f32"; do
  for r in 1 2; do
    for e in "${A:-_A=0}" "${B:-_B=0}"; do
      echo -n "$e $cfg: "
      env $e timeout 160 python bench.py $Q $cfg 2>&1 | grep '^{' | python -c 'import json,sys; d=json.loads(sys.stdin.read()); h=d["host_path"]; print("device", d["value"], "host", h["fps"], "sync", h["fps_synchronous"], "frac", h["frac_of_device_resident"], "1-frame ms", h["single_frame_operator_call_ms"])'
    done
  done
done
