#!/bin/bash
# round 6, job 35: a single 752x480 f32 frame under each switch's other value (are the defaults still the best on this tree?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r35; mkdir -p $out
lat() { python tools/latency_stages.py --calls 300 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['p50_ms'], d['p99_ms'])"; }
for rep in 1 2; do
  for sw in "X=0" "SPFE_F32_HEADS=1" "SPFE_PBTAIL=0" "SPFE_REPLAY_WAVES=8" "SPFE_POOL_SPLIT=0" "SPFE_POOL_SPLIT=1" "SPFE_TILE2_MASK=0" "SPFE_TILE2_MASK=126" "SPFE_SPARSE_DA=0" "SPFE_SPARSE_DB=0" "SPFE_SEL_EXT_EVENT=0" "SPFE_INLINE_CHAIN=0" "SPFE_TILE16X4=0" "X=1"; do
    echo -n "$sw: " >> $out/sweep.txt; env $sw bash -c "$(declare -f lat); lat" >> $out/sweep.txt
  done
done
for rep in 1 2; do
  for sw in "X=0" "SPFE_REPLAY_WAVES=8" "SPFE_SPARSE_DA=0" "SPFE_PBTAIL=4" "SPFE_BF16_WS=15,3" "SPFE_BF16_WS=15,8"; do
    echo -n "bf16 $sw: " >> $out/sweep.txt; env $sw bash -c "$(declare -f lat); lat --precision bf16" >> $out/sweep.txt
  done
done
cat $out/sweep.txt
