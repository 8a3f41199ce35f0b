#!/bin/bash
# round 6, job 27: the final tree: the whole GPU suite, the bench lines and the profiles
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r27; mkdir -p $out
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > $out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.log
tail -5 $out/pytest_gpu.log
bash tools/bench_round.sh r06 > $out/bench_round.log 2>&1
bash tools/profile_round.sh r06 > $out/profile_round.log 2>&1
BENCH_ARGS="--precision bf16 --height 720 --width 1280" ANCHOR=pbtail_bf16 bash tools/microbench/run_steptrace.sh > $out/steptrace_bf16.log 2>&1
tail -3 $out/bench_round.log; ls gpurun_out/round_r06 | head -50
