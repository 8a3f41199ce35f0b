#!/bin/bash
# round 6, job 8: conv1b (fused conv1a) with the producers' u8 patch loads one tile ahead and the tile queue three ahead: bit checks,
# kernel A/B, timing builds, step A/B, bf16 tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r8; mkdir -p $out
B=tools/microbench/bin
for args in "64 96 2 64 1 3 fuse" "120 160 1 64 1 3 fuse" "120 160 1 64 0 3" "24 40 3 64 1 3" "240 376 3 64 1 10 fuse" "16 32 1 64 0 3" "240 376 8 64 0 3" "480 752 2 64 1 5 fuse" "360 640 2 64 1 5"; do
  echo "== w_pre $args" >> $out/probe.txt; timeout 120 $B/w_pre $args 2>&1 | grep -v "sampled" | tail -2 >> $out/probe.txt
done
for rep in 1 2 3; do for v in w_old w_swz w_pre; do
  echo -n "$v 720p fuse: " >> $out/probe.txt
  PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v 720 1280 8 64 1 200 fuse 2>&1 | grep "^conv" | sed 's/.*ws /ws /' >> $out/probe.txt
done; done
for rep in 1 2; do for v in w_old w_pre; do
  for a in "360 640 8 64 0 200" "360 640 8 64 1 200" "480 752 8 64 1 200 fuse" "180 320 8 64 0 200"; do echo -n "$v $a: " >> $out/probe.txt; PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v $a 2>&1 | grep "^conv" | sed 's/.*ws /ws /' >> $out/probe.txt; done
done; done
for v in w_t w_pre_t w_t_a2 w_pre_t_a2; do echo "== $v" >> $out/probe.txt; PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v 720 1280 8 64 1 100 fuse 2>&1 | grep -v "^(fuse" >> $out/probe.txt; done
( time timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_parity.py -x -q -k "bf16" ) > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
ONLY=bf16_720p bash tools/microbench/ab_lib.sh $B/libspfe_prev.so 3 > $out/ab_lib.txt 2>&1
ONLY=bf16_752 bash tools/microbench/ab_lib.sh $B/libspfe_prev.so 2 >> $out/ab_lib.txt 2>&1
cat $out/probe.txt $out/ab_lib.txt; tail -3 $out/pytest.log
