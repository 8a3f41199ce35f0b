cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab3
C="--no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --steps 60 --warmup 10"
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  SPFE_PBTAIL=$v rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ab3/kt_$v -o t -- python $GRAFT_REPO_ROOT/bench.py $C --precision bf16 --height 720 --width 1280 --sync-cov > $GRAFT_REPO_ROOT/gpurun_out/ab3/log_$v.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $GRAFT_REPO_ROOT/gpurun_out/ab3/kt_$v -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/ab3/stats_$v.txt 2>&1
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/ab3/kt_$v
done
cd $GRAFT_REPO_ROOT
grep -E "tail|head1x1_bf16_kernel<65" gpurun_out/ab3/stats_0.txt gpurun_out/ab3/stats_1.txt | cut -c1-200
