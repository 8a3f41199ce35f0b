#!/bin/bash
# After `gpurun -- bash tools/profile_round.sh <tag>` (+ the bench.py runs that write gpurun_out/round_<tag>/bench_*.json):
# copy the summaries that DESIGN.md quotes into profiles/ (tracked) and rebuild the stamped traffic files.
#   usage (build container, repo root): bash tools/collect_round_profiles.sh <tag> <prefix>     e.g.  r02 r02
set -e
tag=$1; pre=$2
R=gpurun_out/round_$tag
declare -A ARGS=([f32_async]="" [f32_sync]="--sync-cov" [f32_sync_nosplit]="--sync-cov  (SPFE_SPLIT=0: one launch per layer)" [bf16_720p_async]="--precision bf16 --height 720 --width 1280" [bf16_720p_sync]="--precision bf16 --height 720 --width 1280 --sync-cov" [bf16_752_async]="--precision bf16")
for f in f32_async f32_sync f32_sync_nosplit bf16_720p_async bf16_720p_sync bf16_752_async; do
  {
    echo "# profiles/${pre}_kernel_stats_$f.txt — rocprofv3 --kernel-trace --stats of: python bench.py --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --steps 100 --warmup 10 ${ARGS[$f]}  (tools/profile_round.sh)"
    echo "# bench line of the same run: $(python -c 'import json,sys; d=json.load(open(sys.argv[1])); print("value", d["value"], "frames/s, ms_per_step", d["ms_per_step"], ", conv1b kernel_ms from the in-region HIP events", d["roofline"]["kernel_ms"], ", frac", d["roofline"]["frac"])' $R/bench_under_trace_$f.json)"
    sed -n 2,40p $R/kernel_stats_$f.txt | cut -c1-175
  } > profiles/${pre}_kernel_stats_$f.txt
done
for n in f32 bf16_720p bf16_752 bf16_720p_rw0; do
  {
    echo "# profiles/${pre}_pmc_$n.txt — rocprofv3 PMC passes (tools/profile_round.sh: separate runs, --kernel-trace --pmc <ctrs> only; bench.py --steps 2 --warmup 1 --sync-cov)"
    echo "# rows: kernel, counter, sum over dispatches, dispatches x instances, min, max per dispatch-instance (SQ_*: one per shader engine, x32 for the chip; GRBM: one per XCD)"
    grep -E "conv_f32_kernel<1,64|conv_bf16_ws_kernel<true,2>|conv1a|cov_replay|conv_bf16_rw_kernel|conv_bf16_kernel<128" $R/pmc_summary_$n.txt | grep -v "^#" | grep -E "FETCH|WRITE|SQ_|GRBM|TCC|LDS" | cut -c1-175
  } > profiles/${pre}_pmc_$n.txt
done
if [ -f $R/kernel_stats_f32_batch1.txt ]; then
  { echo "# profiles/${pre}_kernel_stats_f32_batch1.txt — rocprofv3 --kernel-trace --stats of: python tools/latency_stages.py --calls 200  (single 752x480 frames, synchronous calls; two passes: without and with per-stage events)"; echo "# $(cat $R/latency_stages_f32_batch1.json)"; sed -n 2,40p $R/kernel_stats_f32_batch1.txt | cut -c1-175; } > profiles/${pre}_kernel_stats_f32_batch1.txt
  { echo "# profiles/${pre}_timeline_f32_batch1.txt — one single-frame call from the same trace (tools/rocpd_timeline.py): start offset, duration, gap to the latest earlier end, hardware queue"; cat $R/timeline_f32_batch1.txt; } > profiles/${pre}_timeline_f32_batch1.txt
fi
[ -f $R/probes.txt ] && { echo "# profiles/${pre}_probes.txt — tools/microbench probes run on the GPU box by tools/profile_round.sh (clock_probe: MFMAs only, operands in registers: the clock ceiling; conv_rw_plain: conv_bf16_rw.hip stand-alone on random data; mfma_chain_probe: dependent f32 MFMA chains, 16x16x4 against the fmaf chain)"; cat $R/probes.txt; } > profiles/${pre}_probes.txt
for f in $R/bench_*.json; do b=$(basename $f); case $b in bench_under_trace*) ;; *) cp $f profiles/${pre}_$b;; esac; done
python tools/make_traffic_json.py profiles/${pre}_pmc_f32.txt "conv_f32_kernel<1,64,3,16,4,1,4,2" 480 752 8 conv_f32.hip share=0.93333 > profiles/conv1b_traffic.json
python tools/make_traffic_json.py profiles/${pre}_pmc_bf16_752.txt "conv_bf16_ws_kernel<true,2>" 480 752 8 conv_bf16_ws.hip conv1a_mfma.h > profiles/conv1b_bf16_traffic.json
python tools/make_traffic_json.py profiles/${pre}_pmc_bf16_720p.txt "conv_bf16_ws_kernel<true,2>" 720 1280 8 conv_bf16_ws.hip conv1a_mfma.h > profiles/conv1b_bf16_720p_traffic.json
ls profiles | grep "^$pre" | wc -l
