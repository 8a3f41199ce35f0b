for d in 1 0 1 0; do
  echo "== dyn $d"
  SPFE_BF16_DYN_QUEUE=$d python bench.py --precision bf16 --height 720 --width 1280 --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-aten 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('parity_frame0'), {k:v for k,v in d.get('stage_ms',{}).items() if k in ('conv3b','conv4a','conv4b','convPaDa')})"
done
for d in 1 0; do
  SPFE_BF16_DYN_QUEUE=$d python bench.py --precision bf16 --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-aten 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('752', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v for k,v in d.get('stage_ms',{}).items() if k in ('conv3b','conv4a','conv4b','convPaDa')})"
done
