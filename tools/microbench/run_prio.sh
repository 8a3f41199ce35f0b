for n in 0 16 32 64 0 32; do
  SPFE_SIDE_CUS=$n python bench.py --precision bf16 --height 720 --width 1280 --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --no-aten 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('720p side_cus=$n', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
for n in 0 32 64; do
  SPFE_SIDE_CUS=$n python bench.py --precision bf16 --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --no-aten 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('752 side_cus=$n', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  SPFE_SIDE_CUS=$n python bench.py --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --no-aten 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 side_cus=$n', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
