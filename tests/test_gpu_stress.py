"""GPU: the randomised pipeline stress (tools/stress_pipeline.py) as a test — pipelined device calls and the pipelined host
path against synchronous calls over random sizes, batch sizes, frames per call, feature counts, precisions, detectors, with two
side chains in flight forced on / off / by workload and generation-code wraps of the covariance maps every few calls: records
bit-identical."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cases,seed", [(16, 1), (10, 11)])
def test_pipeline_stress(cases, seed):
    env = {k: v for k, v in os.environ.items() if not k.startswith("SPFE_")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_pipeline.py"), str(cases), str(seed)], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "all records bit-identical" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("cases,seed", [(10, 3)])
def test_sync_parts_stress(cases, seed):
    """tools/stress_sync_parts.py: the synchronous host calls in every form the drop-in uses them (whole, in parts with early maps
    and rows, the caller's map buffers swapped between calls, lazy heat_inv, early copies on / off) interleaved on one handle."""
    env = {k: v for k, v in os.environ.items() if not k.startswith("SPFE_")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_sync_parts.py"), str(cases), str(seed)], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "all records and maps bit-identical" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
