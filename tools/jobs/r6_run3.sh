#!/bin/bash
# round 6, job 3: conv1a store forms A/B (old 8 x b64 / 4 x b128 whole pieces / 8 x b64 conflict-free), all with the packed finish8 except old
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3; mkdir -p $out
B=tools/microbench/bin
for rep in 1 2 3; do for v in w_old w_b128 w_b64cf; do
  echo -n "$v 720p fuse: " >> $out/probe.txt
  PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v 720 1280 8 64 1 200 fuse 2>&1 | grep "^conv" | sed 's/.*ws /ws /' >> $out/probe.txt
done; done
for v in w_old w_b128; do echo -n "$v 752 fuse: " >> $out/probe.txt; PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v 480 752 8 64 1 200 fuse 2>&1 | grep "^conv" | sed 's/.*ws /ws /' >> $out/probe.txt; done
for args in "64 96 2 64 1 3 fuse" "120 160 1 64 1 3 fuse" "240 376 3 64 1 10 fuse"; do
  for v in w_b128 w_b64cf; do echo "== $v $args" >> $out/probe.txt; timeout 120 $B/$v $args 2>&1 | grep -v "sampled" | tail -2 >> $out/probe.txt; done
done
( time timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q ) > $out/pytest.log 2>&1
ONLY=bf16_720p bash tools/microbench/ab_lib.sh $B/libspfe_prev.so 3 > $out/ab_lib.txt 2>&1
ONLY=bf16_752 bash tools/microbench/ab_lib.sh $B/libspfe_prev.so 2 >> $out/ab_lib.txt 2>&1
cat $out/probe.txt; cat $out/ab_lib.txt; tail -3 $out/pytest.log
