cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab4
C="--no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --steps 40 --warmup 5"
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 4 8 6 14 15; do
  SPFE_PBTAIL_DBG=$v rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ab4/kt_$v -o t -- python $GRAFT_REPO_ROOT/bench.py $C --precision bf16 --height 720 --width 1280 --sync-cov > $GRAFT_REPO_ROOT/gpurun_out/ab4/log_$v.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $GRAFT_REPO_ROOT/gpurun_out/ab4/kt_$v -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/ab4/stats_$v.txt 2>&1
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/ab4/kt_$v
  echo -n "dbg=$v: "; grep -E "pbtail" $GRAFT_REPO_ROOT/gpurun_out/ab4/stats_$v.txt | cut -c60-130
done
