#!/bin/bash
# round 6, job 26: do the HBM-bound bf16 layers (conv2a ... convPa at 1280x720) run faster per frame when a call's tensors fit
# the 256 MB Infinity Cache?  Kernel traces of synchronous steps of 1 / 2 / 4 / 8 frames
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r26; mkdir -p $out
common="--no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --precision bf16 --height 720 --width 1280 --sync-cov"
for n in 1 2 4 8; do
  rocprofv3 --kernel-trace --stats -d $out/kt_$n -o trace -- python bench.py $common --steps 60 --warmup 10 --frames-per-gpu $n > $out/kt_$n.log 2>&1
  python tools/rocpd_summary.py $out/kt_$n/*.db > $out/kernel_stats_$n.txt 2>&1
  grep '^{' $out/kt_$n.log | tail -1 | head -c 400 > $out/line_$n.txt
  rm -rf $out/kt_$n
done
for n in 1 2 4 8; do echo "== $n frames"; cut -c1-130 $out/kernel_stats_$n.txt | sed -n 4,16p; done
