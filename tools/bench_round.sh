#!/bin/bash
# The bench lines DESIGN.md / README.md quote, one JSON each (GPU box, via gpurun): usage tools/bench_round.sh <tag>
# -> gpurun_out/round_<tag>/bench_<name>.json ; tools/collect_round_profiles.sh copies them to profiles/<prefix>_bench_<name>.json
tag=$1
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/round_$tag
mkdir -p $out
run() { n=$1; shift; timeout 600 python bench.py "$@" > $out/bench_$n.json 2> $out/bench_$n.err || echo "bench $n failed"; tail -c 300 $out/bench_$n.json | head -c 300; echo; }
run default
run bf16_720p --precision bf16 --height 720 --width 1280 --no-bf16-leg
run bf16_752 --precision bf16 --no-bf16-leg
run f32_720p --height 720 --width 1280 --no-bf16-leg --no-match --no-host-path
run f32_640 --height 480 --width 640 --no-bf16-leg --no-match --no-host-path
run f32_sparse --detector sparse --no-bf16-leg --no-match --no-host-path --no-cpu-baseline
run bf16_720p_sparse --detector sparse --precision bf16 --height 720 --width 1280 --no-bf16-leg --no-match --no-host-path --no-cpu-baseline
