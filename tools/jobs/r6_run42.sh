#!/bin/bash
# round 6, job 42: two side chains against one, second box: bf16 1280x720 x 8 and 3840x2160 x 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r42; mkdir -p $out
C="--no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --warmup 20 --precision bf16"
run() { python bench.py $C "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['frac'], d['roofline'].get('mfma_sustained_tflops'))"; }
for rep in 1 2 3; do for tc in -1 0; do
  echo -n "720p two_chains $tc: " >> $out/ab.txt; SPFE_TWO_CHAINS=$tc run --steps 200 --height 720 --width 1280 >> $out/ab.txt
done; done
for rep in 1 2; do for tc in -1 0; do
  echo -n "2160p b1 two_chains $tc: " >> $out/ab.txt; SPFE_TWO_CHAINS=$tc run --steps 60 --height 2160 --width 3840 --frames-per-gpu 1 >> $out/ab.txt
  echo -n "1080p b8 two_chains $tc: " >> $out/ab.txt; SPFE_TWO_CHAINS=$tc run --steps 100 --height 1080 --width 1920 --frames-per-gpu 8 >> $out/ab.txt
done; done
cat $out/ab.txt
