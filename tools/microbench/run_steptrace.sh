cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/st && mkdir -p gpurun_out/st
rocprofv3 --kernel-trace -d gpurun_out/st/kt -o trace -- python bench.py ${BENCH_ARGS} --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --no-aten --steps 12 > gpurun_out/st/log.txt 2>&1
python tools/rocpd_timeline.py $(find gpurun_out/st/kt -name "*.db" | head -1) ${ANCHOR:-conv1a} -4 > gpurun_out/st/timeline.txt 2>&1
rm -rf gpurun_out/st/kt
cat gpurun_out/st/timeline.txt
