#!/bin/bash
# round 6, job 2: the tests job 1 did not reach + the conv1a piece-layout change (bf16 correctness, kernel A/B, step A/B, LDS conflict counters)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2; mkdir -p $out
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_frontend_chain.py tests/test_gpu_bf16.py -x -q -k "wrap or dropin or adaptor or lazy or captured or bf16" ) > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
B=tools/microbench/bin
for rep in 1 2 3; do for v in w_old w_new; do
  echo -n "$v 720p fuse: " >> $out/probe.txt
  PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v 720 1280 8 64 1 200 fuse 2>&1 | grep "^conv" >> $out/probe.txt
done; done
for args in "64 96 2 64 1 3 fuse" "120 160 1 64 1 3 fuse" "240 376 3 64 1 10 fuse" "480 752 2 64 1 5 fuse"; do
  echo "== w_new $args" >> $out/probe.txt
  timeout 120 $B/w_new $args 2>&1 | grep -v "sampled" >> $out/probe.txt
done
ONLY=bf16_720p bash tools/microbench/ab_lib.sh $B/libspfe_prev.so 3 > $out/ab_lib.txt 2>&1
# LDS conflict counters of the new build, bf16 720p
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $out/pmc -o pmc -- python bench.py --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --steps 2 --warmup 1 --sync-cov --precision bf16 --height 720 --width 1280 > $out/pmc.log 2>&1
python tools/rocpd_summary.py $out/pmc/*.db > $out/pmc_summary.txt 2>&1
rm -rf $out/pmc
python tools/cov_chain_stats.py 480 752 1000 dense 200,201,202,203 > $out/chains.txt 2>&1
tail -3 $out/pytest.log; cat $out/probe.txt | head -30; cat $out/ab_lib.txt
