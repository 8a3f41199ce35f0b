for m in "" "--height 720 --width 1280" "--sync-cov" "--sync-cov --height 720 --width 1280"; do
  python bench.py --precision bf16 $m --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-aten --no-stage-table --latency-calls 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['ms_per_step'], d['latency_batch1_ms']['p50'], d['roofline']['frac'])"
done
