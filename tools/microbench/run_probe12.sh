#!/bin/bash
# wall-clock A/B of non-instrumented probe builds (PROBE_FAST: no CPU comparison); correctness on small shapes with probe_c1
B=tools/microbench/bin
for args in "720 1280 8 64 1 200 fuse" "720 1280 8 64 1 200" "360 640 8 64 0 200" "180 320 8 128 0 200"; do
for v in w_new w_new2 w_new w_new2; do
  echo -n "$v $args: "
  PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v $args 2>&1 | grep "^conv" | sed 's/.*ws /ws /'
done
done
for args in "64 96 2 64 1 3 fuse" "120 160 1 64 1 3 fuse" "120 160 1 64 0 3" "24 40 3 64 1 3" "240 376 3 64 1 10 fuse" "120 188 2 128 0 3" "16 32 1 64 0 3" "240 376 8 64 0 3" "120 188 8 128 1 3"; do
  echo "== probe_c1 $args"
  timeout 120 $B/probe_c1 $args 2>&1 | grep -v "sampled\|^(fuse\|^conv"
done
