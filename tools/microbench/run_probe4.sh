#!/bin/bash
B=tools/microbench/bin
out=gpurun_out/probe_ws4.txt
: > $out
for v in probe_t probe_a1 probe_a3 probe_a4 probe_a5; do
  for args in "720 1280 8 64 1 300"; do
    echo "== $v $args" >> $out
    PROBE_ONLY=new timeout 120 $B/$v $args 2>&1 | grep -v "PROBE\|sampled" >> $out
  done
done
cat $out
