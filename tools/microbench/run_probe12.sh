#!/bin/bash
# Wall-clock A/B of NON-instrumented probe builds + bit-identity on small / ragged shapes.
# Build the variants to compare as tools/microbench/bin/w_<name> (e.g. -DWS_EPI_MICRO=0 vs 1), list them in VARIANTS;
# bin/probe_c1 = the default build, used for the correctness sweep.  PROBE_FAST=1: timing only (no CPU comparison).
B=tools/microbench/bin
VARIANTS=${VARIANTS:-"probe_c1"}
for args in "720 1280 8 64 1 200 fuse" "720 1280 8 64 1 200" "360 640 8 64 0 200" "180 320 8 128 0 200"; do
for v in $VARIANTS $VARIANTS; do
  echo -n "$v $args: "
  PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v $args 2>&1 | grep "^conv" | sed 's/.*ws /ws /'
done
done
for args in "64 96 2 64 1 3 fuse" "120 160 1 64 1 3 fuse" "120 160 1 64 0 3" "24 40 3 64 1 3" "240 376 3 64 1 10 fuse" "120 188 2 128 0 3" "16 32 1 64 0 3" "240 376 8 64 0 3" "120 188 8 128 1 3"; do
  echo "== probe_c1 $args"
  timeout 120 $B/probe_c1 $args 2>&1 | grep -v "sampled\|^(fuse\|^conv"
done
