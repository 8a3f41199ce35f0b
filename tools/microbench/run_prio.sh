cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/hf && mkdir -p gpurun_out/hf
for fh in 1 0; do
SPFE_F32_HEADS=$fh rocprofv3 --kernel-trace --stats -d gpurun_out/hf/kt -o trace -- python bench.py --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --no-aten --steps 6 --sync-cov > gpurun_out/hf/log.txt 2>&1
python tools/rocpd_summary.py gpurun_out/hf/kt/*.db > gpurun_out/hf/stats_$fh.txt 2>&1
grep -i "head\|256,1,64" gpurun_out/hf/stats_$fh.txt | cut -c1-150
rm -rf gpurun_out/hf/kt
done
SPFE_F32_HEADS=1 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed"
