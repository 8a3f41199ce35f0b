#!/bin/bash
# wall-clock A/B of non-instrumented probe builds (PROBE_FAST: no CPU comparison)
B=tools/microbench/bin
for args in "720 1280 8 64 1 200 fuse" "720 1280 8 64 1 200" "360 640 8 64 0 200"; do
for v in w_base w_rs1 w_p3 w_p0 w_base w_rs1 w_p3 w_p0; do
  echo -n "$v $args: "
  PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v $args 2>&1 | grep "^conv" | sed 's/.*ws /ws /'
done
done
