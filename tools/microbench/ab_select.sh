cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04e
C="--no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --steps 200 --warmup 20"
SPFE_SELECT_LEAN=1 timeout 600 python -m pytest tests/test_gpu_selection.py tests/test_gpu_parity.py tests/test_gpu_random_sweep.py tests/test_gpu_sparse_db.py -x -q > gpurun_out/r04e/pytest_lean.log 2>&1; grep -E "passed|failed" gpurun_out/r04e/pytest_lean.log | tail -2
for rep in 1 2; do
for lean in 0 -1; do
  for cfg in "f32:" "bf16_720p:--precision bf16 --height 720 --width 1280" "bf16_752:--precision bf16"; do
    n=${cfg%%:*}; a=${cfg#*:}
    v=$(SPFE_SELECT_LEAN=$lean python bench.py $C $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['frac'])")
    echo "lean=$lean $n $v"
  done
done
done
