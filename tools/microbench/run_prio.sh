for m in 1 0; do
for prec in f32 bf16; do
SPFE_PIPE_COPY_KERNEL=$m python bench.py --precision $prec --no-cpu-baseline --no-bf16-leg --no-match --no-latency --no-stage-table --no-aten 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_path']; print('copy_kernel=$m $prec', 'pipelined', h['fps'], 'sync', h['fps_synchronous'], 'single frame ms', h['single_frame_operator_call_ms'], h['records_ok'])"
done
done
