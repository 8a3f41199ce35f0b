for m in "--height 720 --width 1280" ""; do
  python bench.py --precision bf16 $m --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-aten 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v for k,v in d.get('stage_ms',{}).items() if k in ('convPb','convDb','tail','convPaDa')})"
done
