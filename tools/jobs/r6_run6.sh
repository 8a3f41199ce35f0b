#!/bin/bash
# round 6, job 6: the halo swizzle with the (c & 1) << 2 term (conflict-free b128 halo stores): kernel A/B, bit checks, LDS counters, bf16 tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6; mkdir -p $out
B=tools/microbench/bin
for rep in 1 2 3; do for v in w_old w_b128s w_swz; do
  echo -n "$v 720p fuse: " >> $out/probe.txt
  PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v 720 1280 8 64 1 200 fuse 2>&1 | grep "^conv" | sed 's/.*ws /ws /' >> $out/probe.txt
done; done
for rep in 1 2; do for v in w_old w_swz; do
  for a in "360 640 8 64 0 200" "360 640 8 64 1 200" "480 752 8 64 1 200 fuse"; do echo -n "$v $a: " >> $out/probe.txt; PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v $a 2>&1 | grep "^conv" | sed 's/.*ws /ws /' >> $out/probe.txt; done
done; done
for args in "64 96 2 64 1 3 fuse" "120 160 1 64 1 3 fuse" "120 160 1 64 0 3" "24 40 3 64 1 3" "240 376 3 64 1 10 fuse" "16 32 1 64 0 3" "240 376 8 64 0 3"; do
  echo "== w_swz $args" >> $out/probe.txt; timeout 120 $B/w_swz $args 2>&1 | grep -v "sampled" | tail -2 >> $out/probe.txt
done
( time timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_parity.py -x -q -k "bf16 or lazy or heat_maps_sent" ) > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $out/pmc -o pmc -- python bench.py --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --steps 2 --warmup 1 --sync-cov --precision bf16 --height 720 --width 1280 > $out/pmc.log 2>&1
python tools/rocpd_summary.py $out/pmc/*.db > $out/pmc_summary.txt 2>&1
rm -rf $out/pmc
ONLY=bf16_720p bash tools/microbench/ab_lib.sh $B/libspfe_prev.so 3 > $out/ab_lib.txt 2>&1
ONLY=bf16_752 bash tools/microbench/ab_lib.sh $B/libspfe_prev.so 2 >> $out/ab_lib.txt 2>&1
cat $out/probe.txt $out/ab_lib.txt; tail -3 $out/pytest.log; grep "conv_bf16_ws_kernel<true,2>" $out/pmc_summary.txt
