#!/bin/bash
# The scaling curve of BASELINE.json's metric on ONE node: N = 1, 2, 4, 8 GPUs (those the node has), the same per-GPU
# workload at every N (8 frames of 752x480 per GPU and step = configs[2]'s shard: weak scaling; N = 8 is the 64-frame
# global batch), one JSON line per N.  bench.py launches its own ranks (torch.distributed.run, one per GPU) and exits
# non-zero — no line — when the node has fewer GPUs than N or RCCL's communicator reports another rank count, so a
# line in the output IS an N-GPU measurement.  The N > 1 lines carry `parity_gathered`, `allgather_ms`, `rccl_ranks`,
# `comm_stream_ab` (gather on the library's side stream vs a stream of its own) and `host_alt`.
#   usage: tools/scale_sweep.sh [outdir] [extra bench.py flags ...]      e.g.  tools/scale_sweep.sh gpurun_out/scale --precision bf16
#   env: STEPS / WARMUP (default 20 / 5), MAXN (largest N tried), SPFE_BENCH_BACKEND=gloo (dry run: ranks share the GPUs there are)
# N = 1 runs the driver's command line (--steps 20 --warmup 5), so its `value` is BENCH_rNN.json's.
out=${1:-gpurun_out/scale}; shift
cd "$(dirname "$0")/.." && mkdir -p "$out"
ngpu=$(python -c 'import torch; print(torch.cuda.device_count())' 2>/dev/null || echo 0)
echo "scale_sweep: $ngpu GPU(s) visible" >&2
: > "$out/scale.jsonl"
for n in 1 2 4 8; do
  if [ -n "$MAXN" ] && [ "$n" -gt "$MAXN" ]; then continue; fi
  if [ "$n" -gt "$ngpu" ] && [ -z "$SPFE_BENCH_BACKEND" ]; then echo "scale_sweep: skipping N=$n (only $ngpu GPUs)" >&2; continue; fi
  extra=""; [ "$n" -eq 1 ] && extra="--no-bf16-leg --no-match --no-host-path --no-latency --no-aten"
  python bench.py --gpus "$n" --steps "${STEPS:-20}" --warmup "${WARMUP:-5}" $extra "$@" > "$out/bench_n$n.json" 2> "$out/bench_n$n.err"
  rc=$?
  if [ $rc -ne 0 ] || ! grep -q '^{' "$out/bench_n$n.json"; then echo "scale_sweep: N=$n FAILED (rc $rc), see $out/bench_n$n.err" >&2; continue; fi
  grep '^{' "$out/bench_n$n.json" | tail -1 >> "$out/scale.jsonl"
done
python - "$out/scale.jsonl" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
base = next((r["value"] for r in rows if r["n_gpus"] == 1), None)
for r in rows:
    ab = r.get("comm_stream_ab") or {}
    print("N=%d  %9.1f frames/s  %.3f ms/step  x%.2f of N=1  rccl_ranks=%s  allgather_ms=%s  own_stream=%s side_stream=%s" % (
        r["n_gpus"], r["value"], r["ms_per_step"], r["value"] / base if base else float("nan"),
        (r.get("rccl_ranks") or {}).get("library"), r.get("allgather_ms"),
        (ab.get("own_stream") or {}).get("value"), (ab.get("side_stream") or {}).get("value")))
PY
