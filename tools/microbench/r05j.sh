mkdir -p gpurun_out/r05j; O=gpurun_out/r05j
F="--no-cpu-baseline --no-host-path --no-match --no-latency --no-stage-table --no-aten --no-uhd-leg"
ex() { python -c "
import json,sys; d=json.loads([l for l in open('$1') if l.startswith('{')][-1])
print('$1', d['value'], {k:d[k]['value'] for k in d if isinstance(d[k],dict) and 'value' in d[k] and k.endswith('b8')})"; }
python bench.py $F > $O/a.json 2>/dev/null; ex $O/a.json
GPU_MAX_HW_QUEUES=8 python bench.py $F > $O/b.json 2>/dev/null; ex $O/b.json
python bench.py $F > $O/c.json 2>/dev/null; ex $O/c.json
GPU_MAX_HW_QUEUES=8 python bench.py $F > $O/d.json 2>/dev/null; ex $O/d.json
python bench.py $F --precision bf16 --height 720 --width 1280 --no-bf16-leg --steps 200 --warmup 20 > $O/e.json 2>/dev/null; ex $O/e.json
GPU_MAX_HW_QUEUES=8 python bench.py $F --precision bf16 --height 720 --width 1280 --no-bf16-leg --steps 200 --warmup 20 > $O/f.json 2>/dev/null; ex $O/f.json
