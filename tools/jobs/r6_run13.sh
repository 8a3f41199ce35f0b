#!/bin/bash
# round 6, job 13: with the producers at ~270 VALU per tile, where does the consumers' epilogue / fragment-read placement want to be (TAG 2)?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r13; mkdir -p $out
B=tools/microbench/bin
for rep in 1 2 3; do for v in w_f w_f_m1 w_f_m1s0 w_f_s1 w_f_g1 w_f_g3; do
  echo -n "$v 720p fuse: " >> $out/probe.txt
  PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v 720 1280 8 64 1 200 fuse 2>&1 | grep "^conv" | sed 's/.*ws /ws /' >> $out/probe.txt
done; done
for v in w_f_m1 w_f_s1; do echo "== $v" >> $out/probe.txt; timeout 120 $B/$v 240 376 3 64 1 10 fuse 2>&1 | grep -v sampled | tail -2 >> $out/probe.txt; done
cat $out/probe.txt
