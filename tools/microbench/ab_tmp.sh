for cfg in "bf16 720 1280" "f32 480 752"; do
for v in 0 32 64 96 128 0; do
set -- $cfg $v
echo -n "SIDE_CUS=$4 $cfg: "
SPFE_SIDE_CUS=$4 timeout 300 python bench.py --steps 100 --warmup 10 --precision $1 --height $2 --width $3 --no-cpu-baseline --no-match --no-bf16-leg --no-aten --no-host-path --no-stage-table --no-latency 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
