for cfg in "480 640" "720 1280" "480 752"; do
for v in 0 1 0 1; do
set -- $cfg $v
echo -n "TILE16X4=$3 $1x$2: "
SPFE_TILE16X4=$3 timeout 300 python bench.py --steps 100 --warmup 10 --height $1 --width $2 --no-cpu-baseline --no-match --no-bf16-leg --no-aten --no-host-path --no-stage-table --no-latency 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d.get('parity_frame0'))"
done; done
