#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r14; mkdir -p $out
B=tools/microbench/bin
for rep in 1 2 3; do for v in w_f w_g w_g_sp w_g_pk w_g_sp_pk; do
  echo -n "$v 720p fuse: " >> $out/probe.txt
  PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v 720 1280 8 64 1 200 fuse 2>&1 | grep "^conv" | sed 's/.*ws /ws /' >> $out/probe.txt
done; done
for v in w_g_sp_pk w_g_pk; do for a in "240 376 3 64 1 10 fuse" "120 160 1 64 0 3" "24 40 3 64 1 3"; do echo "== $v $a" >> $out/probe.txt; timeout 120 $B/$v $a 2>&1 | grep -v sampled | tail -2 >> $out/probe.txt; done; done
cat $out/probe.txt
