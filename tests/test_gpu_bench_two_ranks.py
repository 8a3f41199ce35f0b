"""GPU: the N > 1 flow of bench.py as the driver launches it (torch.distributed.run, one process per
rank), dry-run on ONE GPU: both ranks share the device and the record all-gather goes through gloo
(SPFE_BENCH_BACKEND=gloo) instead of RCCL.  Covers the rendezvous, barriers, the communication-stream
all-gather of the record buffers, decoding the last rank's frames and the single JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_gpu():
    env = dict(os.environ, SPFE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--height", "240", "--width", "320",
           "--frames-per-gpu", "3", "--num-features", "200",
           "--no-cpu-baseline", "--no-match", "--no-latency", "--no-stage-table"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["frames_per_gpu"] == 3 and d["config"]["parallelism"] == "dp2"
