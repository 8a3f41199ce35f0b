"""GPU: the N > 1 flow of bench.py as the driver launches it (torch.distributed.run, one process per
rank), dry-run on ONE GPU: both ranks share the device and the record all-gather goes through gloo
(SPFE_BENCH_BACKEND=gloo) instead of RCCL.  Covers the rendezvous, barriers, the communication-stream
all-gather of the record buffers, decoding the last rank's frames, the self-verification fields of the N > 1 line
(parity_gathered: a frame computed by the OTHER rank, as it arrived, against the oracle; allgather_ms; rccl_ranks;
host_alt), the barrier before teardown and the single JSON line."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


SMALL = ["--steps", "3", "--warmup", "1", "--height", "240", "--width", "320", "--frames-per-gpu", "3", "--num-features", "200",
         "--no-cpu-baseline", "--no-match", "--no-latency", "--no-stage-table"]


def _clean_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    return env


def test_plain_python_bench_gpus_2_launches_two_ranks_itself():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run used to read WORLD_SIZE = 1, run on one
    GPU and print n_gpus 1 with rc 0.  It now re-executes itself under torch.distributed.run (gloo dry run: two ranks, one GPU)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, cwd=ROOT,
                         env=_clean_env(SPFE_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["parity_gathered"] is True
    assert "launching 2 ranks" in out.stderr


def test_bench_gpus_2_on_a_one_gpu_box_fails_loudly():
    """... and without the dry-run flag, on a box with fewer GPUs than ranks, it must not print a line at all."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, cwd=ROOT,
                         env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "only 1 device(s) visible" in out.stderr
    # a rendezvous that disagrees with --gpus is refused as well (WORLD_SIZE = 1, --gpus 2)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, cwd=ROOT,
                         env=dict(_clean_env(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_two_ranks_one_gpu():
    env = dict(os.environ, SPFE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"),
           "--gpus", "2"] + SMALL
    t0 = time.time()
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["frames_per_gpu"] == 3 and d["config"]["parallelism"] == "dp2"
    # the line proves itself: rank 1's first frame (global frame 3), rank 0's own and the last one, through the gather
    assert d["parity_gathered"] is True
    assert set(d["parity_gathered_detail"]) == {"frame_3_from_rank_1", "frame_0_from_rank_0", "frame_5_from_rank_1"}
    assert all(v["keypoints_exact"] and v["desc_bitwise"] and v["cov2_bitwise"] for v in d["parity_gathered_detail"].values())
    assert d["allgather_ms"] > 0 and d["allgather"]["bytes_per_rank"] > 0
    assert d["rccl_ranks"]["process_group"] == 2          # (gloo dry run: the library communicator is not in use)
    assert d["host_alt"]["value"] > 0 and d["host_alt"]["records_ok"] and d["host_alt_ms"] > 0
    # teardown: both ranks left through the final barriers (no rank exits while rank 0 is still printing)
    assert "Traceback" not in out.stderr


def test_bench_eight_ranks_one_gpu_configs2_dry_run():
    """BASELINE configs[2] literally, as a dry run: `python bench.py --gpus 8` — 64 frames, 8 per rank, 8 ranks
    that share the one GPU, the record all-gather through gloo — on small frames.  Rank 0 checks a gathered frame FROM EVERY RANK
    (each rank's first, and rank 7's last = the end of the 64-record buffer) against the oracle, bit for bit."""
    small8 = ["--steps", "2", "--warmup", "1", "--height", "128", "--width", "160", "--frames-per-gpu", "8", "--num-features", "100",
              "--no-cpu-baseline", "--no-match", "--no-latency", "--no-stage-table"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + small8, cwd=ROOT,
                         env=_clean_env(SPFE_BENCH_BACKEND="gloo", SPFE_LEGS_TIMEOUT="600"), capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["parallelism"] == "dp8" and d["config"]["frames_per_gpu"] == 8 and d["value"] > 0
    assert d["parity_gathered"] is True
    want = {"frame_%d_from_rank_%d" % (8 * r, r) for r in range(8)} | {"frame_63_from_rank_7"}
    assert set(d["parity_gathered_detail"]) == want
    assert all(v["keypoints_exact"] and v["desc_bitwise"] and v["cov2_bitwise"] for v in d["parity_gathered_detail"].values())
    assert d["rccl_ranks"]["process_group"] == 8 and d["allgather"]["bytes_per_rank"] > 0 and d["host_alt"]["records_ok"]
    assert "legs_incomplete" not in d and list(d)[-1] == "configs"


def test_scale_sweep_script_dry_run(tmp_path):
    """tools/scale_sweep.sh (the N = 1, 2, 4, 8 curve of SCALE_rNN.json) on one GPU, MAXN = 8 (gloo dry run: the ranks share the
    GPU), small frames, 8 frames per rank as configs[2] shards them — one JSON line per N in scale.jsonl, the N = 8 line from
    eight ranks with a gathered frame of every rank checked, the summary table printed."""
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_sweep.sh"), str(tmp_path), "--height", "128", "--width", "160",
                          "--frames-per-gpu", "8", "--num-features", "100", "--no-cpu-baseline", "--no-stage-table", "--no-comm-ab"],
                         cwd=ROOT, env=_clean_env(SPFE_BENCH_BACKEND="gloo", MAXN="8", STEPS="2", WARMUP="1", SPFE_LEGS_TIMEOUT="600"),
                         capture_output=True, text=True, timeout=2400)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [json.loads(l) for l in open(tmp_path / "scale.jsonl")]
    assert [r["n_gpus"] for r in rows] == [1, 2, 4, 8]
    for r in rows[1:]:
        assert r["parity_gathered"] is True and r["config"]["parallelism"] == "dp%d" % r["n_gpus"]
        assert len(r["parity_gathered_detail"]) == r["n_gpus"] + 1
    assert all("N=%d" % n in out.stdout for n in (1, 2, 4, 8))


def test_a_rank_failing_inside_a_leg_does_not_cost_the_line():
    """The legs behind the headline (gathered-frame parity, the collective alone, the comm-stream A/B, the host alternative) run
    under a watchdog on every rank: rank 1 fails inside the last leg (injected), rank 0 waits for it in the leg's barrier — and
    the line is still printed once, with what was measured, exit code 0."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, cwd=ROOT,
                         env=_clean_env(SPFE_BENCH_BACKEND="gloo", SPFE_BENCH_FAIL_LEG="host_alt:1", SPFE_LEGS_TIMEOUT="40"),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["parity_gathered"] is True and d["allgather_ms"] > 0
    assert "host_alt" in d["legs_incomplete"] and "host_alt" not in d
    assert "leg host_alt failed" in out.stderr


def test_strict_legs_put_a_failed_leg_into_the_exit_status():
    """SPFE_BENCH_STRICT_LEGS=1: the same injected failure — the line is still printed once, complete as far as it got,
    and the job's exit status is non-zero (the failing rank leaves with 3, two seconds after rank 0 has printed)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, cwd=ROOT,
                         env=_clean_env(SPFE_BENCH_BACKEND="gloo", SPFE_BENCH_FAIL_LEG="host_alt:1", SPFE_LEGS_TIMEOUT="40",
                                        SPFE_BENCH_STRICT_LEGS="1"),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "host_alt" in d["legs_incomplete"]
