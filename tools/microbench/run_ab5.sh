cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab5
python -m pytest tests/test_gpu_bf16.py -x -q > gpurun_out/ab5/pytest_bf16.txt 2>&1
tail -3 gpurun_out/ab5/pytest_bf16.txt
for r in 1 2; do for e in "SPFE_PBTAIL=0" "SPFE_PBTAIL_WAVES=2" "SPFE_PBTAIL_WAVES=4" "SPFE_PBTAIL=1"; do echo -n "$e b1 720p: "; env $e python tools/latency_stages.py --precision bf16 --height 720 --width 1280 --calls 400 2>&1 | grep p50 | cut -c1-40; done; done | tee gpurun_out/ab5/b1.txt
for e in "SPFE_PBTAIL=0" "SPFE_PBTAIL=1"; do echo -n "$e b1 752 bf16: "; env $e python tools/latency_stages.py --precision bf16 --calls 400 2>&1 | grep p50 | cut -c1-40; done | tee -a gpurun_out/ab5/b1.txt
python -m pytest tests -m gpu -x -q > gpurun_out/ab5/pytest_all.txt 2>&1
tail -3 gpurun_out/ab5/pytest_all.txt
