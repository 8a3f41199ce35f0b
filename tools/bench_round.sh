#!/bin/bash
# The bench lines DESIGN.md / README.md quote, one JSON each (GPU box, via gpurun): usage tools/bench_round.sh <tag>
# -> gpurun_out/round_<tag>/bench_<name>.json ; tools/collect_round_profiles.sh copies them to profiles/<prefix>_bench_<name>.json
tag=$1
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/round_$tag
mkdir -p $out
run() { n=$1; shift; timeout 600 python bench.py "$@" > $out/bench_$n.json 2> $out/bench_$n.err || echo "bench $n failed"; tail -c 300 $out/bench_$n.json | head -c 300; echo; }
run default --gpus 1 --steps 20 --warmup 5      # the driver's command line (BENCH_rNN.json): the compact line ...
cp gpurun_out/bench_full.json $out/bench_default_full.json 2>/dev/null   # ... and every leg's full object of the same run (bench.py writes it beside)
L="--verbose-line --steps 200 --warmup 20"                        # the others: steady state (a 20-step region reads up to 8 % low in bf16 mode)
run bf16_720p $L --precision bf16 --height 720 --width 1280 --no-bf16-leg
run bf16_752 $L --precision bf16 --no-bf16-leg
run f32_720p $L --height 720 --width 1280 --no-bf16-leg --no-match --no-host-path
run f32_640 $L --height 480 --width 640 --no-bf16-leg --no-match --no-host-path
run f32_sparse $L --detector sparse --no-bf16-leg --no-match --no-host-path --no-cpu-baseline
run bf16_720p_sparse $L --detector sparse --precision bf16 --height 720 --width 1280 --no-bf16-leg --no-match --no-host-path --no-cpu-baseline
