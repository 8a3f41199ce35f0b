#!/bin/bash
# PMC passes for the bf16 conv kernels (GPU box only; counters in their own runs, --kernel-trace only).
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p $out
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d $out/p$i -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-match --sync-cov --precision bf16 "$@" > $out/p$i.log 2>&1 || echo "pass $i failed"
done
python tools/rocpd_summary.py $out/p*/*.db > $out/summary.txt 2>&1 || true
grep -A400 "PMC counters" $out/summary.txt | grep -E "conv_bf16_kernel<64,true,false>|PMC" | cut -c1-175
