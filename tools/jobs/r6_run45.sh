#!/bin/bash
# round 6, job 45: bf16 752x480 x 8 and f32 1280x720 x 8 under each switch's other value
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r45; mkdir -p $out
C="--no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --warmup 20"
run() { python bench.py $C "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['frac'])"; }
for rep in 1 2; do
  for sw in "X=0" "SPFE_SPLIT=0" "SPFE_TWO_CHAINS=1" "SPFE_REPLAY_WAVES=2" "SPFE_REPLAY_WAVES=8" "SPFE_SPARSE_DA=1" "SPFE_SPARSE_DB=2" "SPFE_PBTAIL=4" "SPFE_PBTAIL=2" "SPFE_BF16_WS=15,3" "SPFE_BF16_WS=1" "SPFE_BF16_RW=0" "SPFE_BF16_DYN_QUEUE=0" "SPFE_DEFER_JOIN=0" "SPFE_TAIL_PER_HALF=0" "X=1"; do
    echo -n "bf16_752 $sw: " >> $out/sweep.txt; env $sw bash -c "$(declare -f run); C='$C'; run --steps 300 --precision bf16" >> $out/sweep.txt
  done
  for sw in "X=0" "SPFE_SPLIT=0" "SPFE_TILE16X4=0" "SPFE_TILE16X4=2" "SPFE_REPLAY_WAVES=8" "SPFE_DEFER_JOIN=0" "X=1"; do
    echo -n "f32_720p $sw: " >> $out/sweep.txt; env $sw bash -c "$(declare -f run); C='$C'; run --steps 60 --height 720 --width 1280" >> $out/sweep.txt
  done
done
cat $out/sweep.txt
