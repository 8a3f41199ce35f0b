#!/bin/bash
# round 6, job 46: the bench lines and the profiles of the last tree (one side chain at 1280x720 x 8, conv1a inside conv1b in single-frame f32 calls)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r46; mkdir -p $out
bash tools/bench_round.sh r06 > $out/bench_round.log 2>&1
bash tools/profile_round.sh r06 > $out/profile_round.log 2>&1
tail -3 $out/bench_round.log | cut -c1-300; ls gpurun_out/round_r06 | wc -l
