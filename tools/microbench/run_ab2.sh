cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab2
python -m pytest tests/test_gpu_bf16.py -x -q > gpurun_out/ab2/pytest_bf16.txt 2>&1
tail -5 gpurun_out/ab2/pytest_bf16.txt
ONLY=bf16_720p bash tools/microbench/ab_env.sh SPFE_PBTAIL "0 1" 3 > gpurun_out/ab2/ab_720p.txt 2>&1
ONLY=bf16_752 bash tools/microbench/ab_env.sh SPFE_PBTAIL "0 1" 2 > gpurun_out/ab2/ab_752.txt 2>&1
cat gpurun_out/ab2/ab_720p.txt gpurun_out/ab2/ab_752.txt
for e in 0 1; do echo -n "PBTAIL=$e b1: "; SPFE_PBTAIL=$e python tools/latency_stages.py --precision bf16 --height 720 --width 1280 --calls 400 2>&1 | grep p50 | cut -c1-60; done | tee gpurun_out/ab2/b1.txt
