#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of one bench invocation.
#   tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/*.db + summary txt
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o trace -- python bench.py "$@" > gpurun_out/prof_$tag/bench.log 2>&1 || true
grep '^{' gpurun_out/prof_$tag/bench.log | tail -1 > gpurun_out/prof_$tag/bench.json || true
python tools/rocpd_summary.py gpurun_out/prof_$tag/*.db > gpurun_out/prof_$tag/kernel_stats.txt
cat gpurun_out/prof_$tag/kernel_stats.txt | cut -c1-175
