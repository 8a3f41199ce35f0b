#!/bin/bash
# round 6, job 9: the tile queue's atomics: one per tile / 4 / 16 tiles per atomic / none (static), kernel times and the timing builds
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r9; mkdir -p $out
B=tools/microbench/bin
for rep in 1 2 3; do for v in w_old w_c1 w_c4 w_c16 w_st; do
  echo -n "$v 720p fuse: " >> $out/probe.txt
  PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v 720 1280 8 64 1 200 fuse 2>&1 | grep "^conv" | sed 's/.*ws /ws /' >> $out/probe.txt
done; done
for rep in 1 2; do for v in w_old w_c4 w_st; do
  for a in "360 640 8 64 0 200" "360 640 8 64 1 200" "480 752 8 64 1 200 fuse" "180 320 8 64 0 200"; do echo -n "$v $a: " >> $out/probe.txt; PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v $a 2>&1 | grep "^conv" | sed 's/.*ws /ws /' >> $out/probe.txt; done
done; done
for v in w_pre_t_a2 w_c4_t_a2 w_pre_t w_c4_t; do echo "== $v" >> $out/probe.txt; PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v 720 1280 8 64 1 100 fuse 2>&1 | grep -v "^(fuse" >> $out/probe.txt; done
for args in "64 96 2 64 1 3 fuse" "120 160 1 64 0 3" "24 40 3 64 1 3" "240 376 3 64 1 10 fuse" "240 376 8 64 0 3" "480 752 2 64 1 5 fuse"; do
  echo "== w_c4 $args" >> $out/probe.txt; timeout 120 $B/w_c4 $args 2>&1 | grep -v "sampled" | tail -2 >> $out/probe.txt
done
ONLY=bf16_720p bash tools/microbench/ab_lib.sh $B/libspfe_prev.so 3 > $out/ab_lib.txt 2>&1
cat $out/probe.txt $out/ab_lib.txt
