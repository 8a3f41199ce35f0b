#!/bin/bash
B=tools/microbench/bin
for args in "64 96 2 64 1 3 fuse" "120 160 1 64 0 3" "24 40 3 64 1 3" "240 376 3 64 1 10 fuse" "120 188 2 128 0 3" "16 32 1 64 0 3" "240 376 8 64 0 3" "120 188 8 128 1 3"; do
  echo "== probe_c1 $args"
  timeout 120 $B/probe_c1 $args 2>&1 | grep -v "sampled\|^(fuse"
done
