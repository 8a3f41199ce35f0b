#!/bin/bash
# round 6, job 41: two side chains (the twin handle) against one, bf16 pipelined, dense and sparse detector, 1280x720 and 1920x1080
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r41; mkdir -p $out
C="--no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --warmup 20 --steps 200 --precision bf16"
run() { python bench.py $C "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['frac'])"; }
for rep in 1 2 3 4; do for tc in -1 0; do
  echo -n "720p two_chains $tc: " >> $out/ab.txt; SPFE_TWO_CHAINS=$tc run --height 720 --width 1280 >> $out/ab.txt
done; done
for rep in 1 2; do for tc in -1 0; do
  echo -n "720p sparse two_chains $tc: " >> $out/ab.txt; SPFE_TWO_CHAINS=$tc run --height 720 --width 1280 --detector sparse >> $out/ab.txt
  echo -n "1080p two_chains $tc: " >> $out/ab.txt; SPFE_TWO_CHAINS=$tc run --height 1080 --width 1920 --frames-per-gpu 4 >> $out/ab.txt
done; done
cat $out/ab.txt
