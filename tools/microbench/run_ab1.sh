cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab1
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
 for e in "SPFE_SEL_EXT_EVENT=0" "SPFE_SEL_EXT_EVENT=1" "SPFE_FUSE_CONV1A=1"; do
  echo -n "$e: " >> gpurun_out/ab1/lat.txt
  env $e python tools/latency_stages.py --calls 600 2>&1 | grep -i "p50" | head -1 >> gpurun_out/ab1/lat.txt
 done
done
python -m pytest tests/test_gpu_parity.py -x -q -k "batch1 or single or sync" > gpurun_out/ab1/pytest.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/ab1/trace -o b1 -- python $GRAFT_REPO_ROOT/tools/latency_stages.py --calls 100 > $GRAFT_REPO_ROOT/gpurun_out/ab1/trace.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/rocpd_timeline.py $(find gpurun_out/ab1/trace -name "*.db" | head -1) conv1a 60 > gpurun_out/ab1/timeline.txt 2>&1 || true
