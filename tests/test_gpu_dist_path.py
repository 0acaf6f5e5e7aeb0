"""The RCCL code path of the benchmark step on ONE GPU: `bench.py --force-dist` initialises the "nccl"
process group with a single rank and runs the step's all_gather_into_tensor / barrier / max-reduce exactly as the N > 1 job
does. Its results must be those of the run without a process group -- the only multi-GPU code that can touch a GPU in this
environment stays exercised while the 8-GPU bench is the driver's to launch."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "8", "--size", "512",
           "--no-extras", "--no-cpu-baseline", *extra]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_force_dist_runs_the_rccl_path_with_identical_results():
    plain = _bench(port=29611)
    dist = _bench("--force-dist", port=29612)
    assert plain["config"]["collective_backend"] is None and plain["config"]["n_ranks_seen"] == 1
    assert dist["config"]["n_ranks_seen"] == 1 and dist["n_gpus"] == 1
    assert dist["config"]["collective_backend"].startswith("nccl")
    assert dist["config"]["status_bits"] == 0 and plain["config"]["status_bits"] == 0
    assert dist["config"]["mean_instances_per_frame"] == plain["config"]["mean_instances_per_frame"] > 0
    # the gathered packed rows (instance peaks, values, scores, counts, status words) of the last step, bit for bit
    assert dist["config"]["result_digest"] == plain["config"]["result_digest"]
