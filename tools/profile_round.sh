#!/bin/bash
# One round's profiles (GPU box, via gpurun): kernel-trace summaries of the bench commands the numbers in
# DESIGN.md come from, and PMC passes (their own runs, --kernel-trace --pmc only) for the dominant kernel of
# each precision.  usage: tools/profile_round.sh <tag>  -> gpurun_out/round_<tag>/...
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/round_$tag
mkdir -p $out
common="--no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table"
trace() {   # name, bench args
  name=$1; shift
  rocprofv3 --kernel-trace --stats -d $out/kt_$name -o trace -- python bench.py $common --steps 100 --warmup 10 "$@" > $out/kt_$name.log 2>&1
  python tools/rocpd_summary.py $out/kt_$name/*.db > $out/kernel_stats_$name.txt 2>&1
  grep '^{' $out/kt_$name.log | tail -1 > $out/bench_under_trace_$name.json
}
trace f32_async
trace f32_sync --sync-cov
# one launch per layer (no two-stream half batches), synchronous: the trace per-layer fractions are recomputed from
SPFE_SPLIT=0 trace f32_sync_nosplit --sync-cov
# a single frame per call (BASELINE configs[1] as written): per-kernel durations and one call's timeline
rocprofv3 --kernel-trace --stats -d $out/kt_b1 -o trace -- python tools/latency_stages.py --calls 200 > $out/kt_b1.log 2>&1
python tools/rocpd_summary.py $out/kt_b1/*.db > $out/kernel_stats_f32_batch1.txt 2>&1
python tools/rocpd_timeline.py $out/kt_b1/*.db "conv_f32_kernel<2," 60 > $out/timeline_f32_batch1.txt 2>&1
grep '^{' $out/kt_b1.log | tail -1 > $out/latency_stages_f32_batch1.json
trace bf16_720p_async --precision bf16 --height 720 --width 1280
trace bf16_720p_sync --precision bf16 --height 720 --width 1280 --sync-cov
trace bf16_752_async --precision bf16
pmc() {   # name, counters, bench args
  name=$1; ctrs=$2; shift; shift
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d $out/pmc_$name -o pmc -- python bench.py $common --steps 2 --warmup 1 --sync-cov "$@" > $out/pmc_$name.log 2>&1 || echo "pmc $name failed"
}
if [ -n "$SKIP_PMC" ]; then rm -rf $out/kt_*/; ls $out | wc -l; exit 0; fi   # SKIP_PMC=1: the traces only
for cfg in "f32:" "bf16_720p:--precision bf16 --height 720 --width 1280" "bf16_752:--precision bf16"; do
  n=${cfg%%:*}; a=${cfg#*:}
  pmc ${n}_fetch "FETCH_SIZE" $a
  pmc ${n}_write "WRITE_SIZE" $a
  pmc ${n}_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU" $a
  pmc ${n}_grbm "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum" $a
  python tools/rocpd_summary.py $out/pmc_${n}_*/*.db > $out/pmc_summary_$n.txt 2>&1
done
# the Cin = 128 layers on the streamed-weight kernel (conv_bf16.hip), for the comparison with conv_bf16_rw.hip above
export SPFE_BF16_RW=0
a="--precision bf16 --height 720 --width 1280"
pmc bf16_720p_rw0_fetch "FETCH_SIZE" $a
pmc bf16_720p_rw0_write "WRITE_SIZE" $a
pmc bf16_720p_rw0_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU" $a
pmc bf16_720p_rw0_grbm "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum" $a
python tools/rocpd_summary.py $out/pmc_bf16_720p_rw0_*/*.db > $out/pmc_summary_bf16_720p_rw0.txt 2>&1
unset SPFE_BF16_RW
# probes DESIGN.md leans on (built by tools/microbench/build_probes.sh in the build container)
for pb in clock_probe conv_rw_plain mfma_chain_probe; do
  [ -x tools/microbench/bin/$pb ] && { echo "== $pb"; if [ $pb = conv_rw_plain ]; then for a in "180 320 8 128 1 4 50" "90 160 8 512 0 4 50" "90 160 8 128 0 4 50" "90 160 8 128 0 2 50"; do tools/microbench/bin/$pb $a; done; else tools/microbench/bin/$pb; fi; } >> $out/probes.txt 2>&1
done
rm -rf $out/kt_*/ $out/pmc_*/   # the databases are large; the summaries are what is kept
ls -la $out
