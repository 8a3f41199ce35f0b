for d in -1 3 2 1 0 -1 3; do
  SPFE_DEFER_SIDE_LAYER=$d python bench.py --precision bf16 --height 720 --width 1280 --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --no-aten 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('720p defer=$d', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('parity_frame0'))"
done
for d in -1 3 1 0; do
  SPFE_DEFER_SIDE_LAYER=$d python bench.py --precision bf16 --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --no-aten 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('752 defer=$d', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
