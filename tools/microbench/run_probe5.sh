#!/bin/bash
P=tools/microbench/bin/conv_ws_probe
for args in "120 160 1 64 1 3" "120 160 1 64 0 3" "128 160 1 64 1 3" "240 320 1 64 1 3" "64 96 1 64 1 3" "16 32 1 64 1 3" "24 40 3 64 1 3" "120 160 2 64 1 3"; do
  timeout 120 $P $args 2>&1 | grep -v "^conv"
done
