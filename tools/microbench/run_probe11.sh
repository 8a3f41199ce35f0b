#!/bin/bash
# timing + ablation builds of the probe on the fused (conv1a inside) and plain conv1b shapes
B=tools/microbench/bin
for v in probe_a0 probe_a2 probe_a4 probe_a5; do
for args in "720 1280 8 64 1 300 fuse" "720 1280 8 64 1 300"; do
  echo "== $v $args"
  PROBE_ONLY=new timeout 120 $B/$v $args 2>&1 | grep -v "sampled\|^(fuse\|PROBE\|mismatch\|max abs"
done
done
