cd "$GRAFT_REPO_ROOT"
C="--no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-stage-table --steps 100 --warmup 10 --precision bf16"
for rep in 1 2; do
for e in "SPFE_SPARSE_DB=0" "SPFE_SPARSE_DB=1 SPFE_SPARSE_DA=1" "SPFE_SPARSE_DB=1 SPFE_SPARSE_DA=0"; do
  for cfg in "--sync-cov" "--sync-cov --frames-per-gpu 2" "--sync-cov --height 480 --width 640" "--sync-cov --height 240 --width 320"; do
    echo -n "[$e] [$cfg] "
    env $e python bench.py $C $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('fps', d['value'], 'b1 p50', d['latency_batch1_ms']['p50'])"
  done
done
done
