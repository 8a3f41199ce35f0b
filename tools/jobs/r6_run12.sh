#!/bin/bash
# round 6, job 12: conv1a of the bf16 mode with the 1/255 folded into the weights and the bias as the MFMA's C operand
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r12; mkdir -p $out
B=tools/microbench/bin
for args in "64 96 2 64 1 3 fuse" "120 160 1 64 1 3 fuse" "24 40 3 64 1 3" "240 376 3 64 1 10 fuse" "480 752 2 64 1 5 fuse" "720 1280 1 64 1 3 fuse"; do
  echo "== w_f $args" >> $out/probe.txt; timeout 120 $B/w_f $args 2>&1 | grep -v "sampled" | tail -2 >> $out/probe.txt
done
for rep in 1 2 3 4; do for v in w_old w_r1 w_f; do
  echo -n "$v 720p fuse: " >> $out/probe.txt
  PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v 720 1280 8 64 1 200 fuse 2>&1 | grep "^conv" | sed 's/.*ws /ws /' >> $out/probe.txt
done; done
for rep in 1 2; do for v in w_old w_f; do
  for a in "480 752 8 64 1 200 fuse" "1080 1920 4 64 1 100 fuse"; do echo -n "$v $a: " >> $out/probe.txt; PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v $a 2>&1 | grep "^conv" | sed 's/.*ws /ws /' >> $out/probe.txt; done
done; done
echo "== w_f_t" >> $out/probe.txt; PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/w_f_t 720 1280 8 64 1 100 fuse 2>&1 | grep -v "^(fuse" >> $out/probe.txt
( time timeout 1800 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_parity.py tests/test_gpu_frontend_chain.py tests/test_gpu_desc_bf16.py -x -q -k "bf16" ) > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
ONLY=bf16_720p bash tools/microbench/ab_lib.sh $B/libspfe_prev.so 3 > $out/ab_lib.txt 2>&1
ONLY=bf16_752 bash tools/microbench/ab_lib.sh $B/libspfe_prev.so 2 >> $out/ab_lib.txt 2>&1
grep -v "^old vs\|PROBE OK" $out/probe.txt; grep -c "PROBE OK" $out/probe.txt; cat $out/ab_lib.txt; tail -5 $out/pytest.log
