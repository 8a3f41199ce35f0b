#!/bin/bash
# Run on the GPU box (via gpurun): PMC counter passes of one short bench run.
# Counters are collected in their own runs with --kernel-trace only (no sys/hip traces).
#   tools/profile_pmc.sh <tag> -> gpurun_out/pmc_<tag>/<pass>/*.db + summary txt
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p $out
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d $out/p$i -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency > $out/p$i.log 2>&1 || echo "pass $i failed"
done
python tools/rocpd_summary.py $out/p*/*.db > $out/summary.txt 2>&1 || true
grep -A200 "PMC counters" $out/summary.txt | grep -E "conv_f32_kernel<64,3,16,4,1,2,2,true|conv1a|PMC" | cut -c1-160
