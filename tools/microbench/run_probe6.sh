#!/bin/bash
B=tools/microbench/bin
out=gpurun_out/probe_ws6.txt
: > $out
for args in "720 1280 8 64 1 200" "480 752 8 64 1 200" "360 640 8 64 0 200" "240 376 8 64 0 100" "120 188 8 128 0 100" "240 376 8 64 1 100" "64 96 2 64 0 5"; do
  for v in probe_c1 probe_c0; do
    echo "== $v $args" >> $out
    timeout 120 $B/$v $args 2>&1 | grep -v "^PROBE OK" >> $out
  done
done
cat $out
