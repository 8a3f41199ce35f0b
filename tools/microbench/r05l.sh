F="--no-cpu-baseline --no-host-path --no-match --no-latency --no-stage-table --no-aten --no-uhd-leg"
for v in 0 -1 0 -1; do SPFE_TWO_CHAINS=$v python bench.py $F 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); x=d['bf16_1280x720_b8']; print('inline seed300 TWO_CHAINS=$v', x['value'], x['roofline']['frac'], d['bf16_752x480_b8']['value'])"; done
ONLY=bf16_720p bash tools/microbench/ab_env.sh SPFE_REPLAY_WAVES "2 8" 2
ONLY=bf16_720p bash tools/microbench/ab_env.sh SPFE_SPARSE_DA "1 2" 2
