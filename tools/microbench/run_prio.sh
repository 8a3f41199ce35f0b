for m in "" "--sync-cov" "" "--sync-cov"; do
  echo "== mode [$m]"
  python bench.py --precision bf16 --height 720 --width 1280 $m --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-aten 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d.get('stage_ms'))"
done
