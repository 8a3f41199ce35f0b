#!/bin/bash
B=tools/microbench/bin
for args in "64 96 2 64 1 5 fuse" "120 160 1 64 1 3 fuse" "240 376 3 64 1 10 fuse" "480 752 8 64 1 100 fuse" "720 1280 8 64 1 100 fuse" "720 1280 8 64 1 100"; do
  timeout 120 $B/probe_c1 $args 2>&1 | grep -v "sampled\|shader clock"
done
