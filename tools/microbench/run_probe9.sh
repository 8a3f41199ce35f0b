#!/bin/bash
B=tools/microbench/bin
for v in probe_c1 probe_a6 probe_a7 probe_a3; do
  for args in "720 1280 8 64 1 200"; do
    echo "== $v $args"
    PROBE_ONLY=new timeout 120 $B/$v $args 2>&1 | grep -v "PROBE\|sampled"
  done
done
