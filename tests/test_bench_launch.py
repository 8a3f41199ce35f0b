"""CPU: `python bench.py --gpus N` launches its own ranks and never degrades to a smaller run.
The compute is not reached here (no GPU): what is checked is the launch and the refusal."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def _run(args, env):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=300)


def test_gpus_2_without_a_rendezvous_launches_two_ranks_and_fails_loudly_without_gpus():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-box test (the GPU form is tests/test_gpu_bench_two_ranks.py)")
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], _env())
    assert "launching 2 ranks" in out.stderr and "--nproc-per-node 2" in out.stderr
    assert out.returncode != 0                                       # torch.distributed.run's exit code comes back
    assert "FATAL: no GPU visible" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]   # no line, not a line with n_gpus 1


def test_a_rendezvous_that_disagrees_with_gpus_is_refused():
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], _env(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert out.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    out = _run(["--gpus", "0"], _env())
    assert out.returncode != 0


def test_numa_binding_reports_and_never_raises():
    """bench.py binds every rank of an N > 1 run to the cores of its GPU's NUMA node; whatever sysfs looks like on the box
    (no such PCI device here), the call reports what it did instead of raising, and leaves the affinity alone when it cannot
    bind."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class _Props:
        pci_domain_id, pci_bus_id, pci_device_id = 0xffff, 0xfe, 0x1f

    class _Torch:
        class cuda:
            @staticmethod
            def get_device_properties(_):
                return _Props()
    before = os.sched_getaffinity(0)
    out = bench.bind_to_gpu_numa(_Torch, 0)
    assert out["bound"] is False and "why" in out
    assert os.sched_getaffinity(0) == before
