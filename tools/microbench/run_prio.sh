cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/hf && mkdir -p gpurun_out/hf
for a in "" "--precision bf16 --height 720 --width 1280"; do
rocprofv3 --kernel-trace --stats -d gpurun_out/hf/kt -o trace -- python bench.py $a --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --no-aten --steps 6 --sync-cov > gpurun_out/hf/log.txt 2>&1
python tools/rocpd_summary.py gpurun_out/hf/kt/*.db > gpurun_out/hf/stats.txt 2>&1
grep -i "select_kernel\|cov_\|heat_norm\|desc_kernel" gpurun_out/hf/stats.txt | cut -c1-125
rm -rf gpurun_out/hf/kt
done
python -m pytest tests/test_gpu_selection.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed"
