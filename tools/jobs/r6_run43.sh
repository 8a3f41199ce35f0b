#!/bin/bash
# round 6, job 43: two side chains against one at smaller calls (where does the twin start to pay?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r43; mkdir -p $out
C="--no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --warmup 20 --precision bf16"
run() { python bench.py $C "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['frac'])"; }
for rep in 1 2; do
  for cfg in "720 1280 6" "720 1280 4" "720 1280 2" "1080 1920 2" "1080 1920 1" "1440 2560 2" "1440 2560 1"; do
    set -- $cfg
    for tc in 1 0; do
      echo -n "$2x$1 b$3 two_chains $tc: " >> $out/ab.txt; SPFE_TWO_CHAINS=$tc run --steps 200 --height $1 --width $2 --frames-per-gpu $3 >> $out/ab.txt
    done
  done
done
cat $out/ab.txt
