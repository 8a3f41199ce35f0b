#!/bin/bash
# Per-kernel times of the post-processing chain (select / replay / descriptors) at the two bench workloads.
# Output under gpurun_out/selprof/.
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/selprof; mkdir -p $O
QUIET="--no-cpu-baseline --no-latency --no-stage-table --no-match --no-bf16-leg --no-aten --no-host-path --sync-cov"
i=0
for cfg in "--precision f32" "--precision bf16 --height 720 --width 1280"; do
  i=$((i+1))
  rm -rf /tmp/selprof_$i; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/selprof_$i -o sel -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 $QUIET $cfg > $O/bench_$i.log 2>&1
  tail -1 $O/bench_$i.log | cut -c1-200
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/selprof_$i -name "*.db" | head -1) > $O/kernel_stats_$i.txt 2>&1
  cut -c1-150 $O/kernel_stats_$i.txt | head -24
done
