"""`python bench.py --gpus N` must start N ranks by itself (the driver's SCALE runs call it that way when they do not wrap it in
torchrun) and must refuse to report an N-GPU line from a different rank count. Checked on CPU with the gloo dry run
(`--dry-run-cpu`: the distributed skeleton of a step -- shard arithmetic, one all-gather of the packed rows, barrier,
max-reduce -- without any GPU work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e,
                          timeout=timeout, cwd=ROOT)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_gpus_2_self_launches_two_ranks_and_gathers_in_rank_order():
    r = _run(["--gpus", "2", "--steps", "3", "--dry-run-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout  # rank 0 only
    # ... and nothing else: library chatter (RCCL's version banner goes to stdout through C stdio) is kept off the descriptor
    # the driver parses (bench.claim_stdout)
    assert [l for l in r.stdout.splitlines() if l.strip()] == [l for l in r.stdout.splitlines() if l.startswith("{")]
    d = lines[0]
    assert d["n_gpus"] == 2 and d["config"]["n_ranks_seen"] == 2 and d["config"]["collective_backend"] == "gloo"
    assert d["rows_in_rank_order"] and d["config"]["global_batch"] == 128
    # both readings of "batch = 64 over N GPUs" are in the ONE line under explicit names: `value` is the
    # weak curve (64 frames per GPU), the configs[3]-literal global batch of 64 rides beside it
    assert d["scaling"] == "weak" and d["value_weak_64_per_gpu"] is not None and d["value_strong_global_batch_64"] is not None
    lit = d["configs3_global_batch_64"]
    assert lit["frames_per_gpu_per_step"] == 32 and lit["global_batch"] == 64 and lit["scaling"] == "strong" and lit["rows_in_rank_order"]


def test_global_batch_is_split_over_the_ranks():
    r = _run(["--gpus", "2", "--steps", "2", "--dry-run-cpu", "--global-batch", "64"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_lines(r.stdout)[0]
    assert d["config"]["frames_per_gpu_per_step"] == 32 and d["config"]["global_batch"] == 64


def test_a_world_size_that_disagrees_with_gpus_is_an_error_not_a_warning():
    # a rank started by some other launcher with WORLD_SIZE=1 while the command line says 2 GPUs
    r = _run(["--gpus", "2", "--steps", "1", "--dry-run-cpu"], env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_more_gpus_than_visible_is_refused():
    # no GPU in the CPU container (and at most 1 on the test box): asking for 64 must fail loudly before anything is launched
    r = _run(["--gpus", "64", "--steps", "1"])
    assert r.returncode != 0 and "visible" in (r.stderr + r.stdout)


def test_literal_configs3_split_8_ranks_of_8_frames():
    """SCALE-day readiness: configs[3] read literally -- a global batch of 64 over the 8 GPUs of a node -- as the driver would
    launch it (`python bench.py --gpus 8 --global-batch 64`), over gloo."""
    r = _run(["--gpus", "8", "--steps", "2", "--dry-run-cpu", "--global-batch", "64"], env={"OMP_NUM_THREADS": "1"}, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    d = lines[0]
    assert d["n_gpus"] == 8 and d["config"]["n_ranks_seen"] == 8 and d["scaling"] == "strong"
    assert d["config"]["frames_per_gpu_per_step"] == 8 and d["config"]["global_batch"] == 64 and d["rows_in_rank_order"]
    assert d["value_strong_global_batch_64"] is not None and d["value_weak_64_per_gpu"] is None


def test_stdout_carries_only_the_json_line_even_when_c_code_prints(tmp_path):
    """claim_stdout(): what C libraries write to file descriptor 1 after it ends up on stderr; emit() reaches the real stdout."""
    code = ("import ctypes, os, sys; sys.path.insert(0, %r); import bench; bench.claim_stdout(); "
            "ctypes.CDLL(None).puts(b'RCCL version : banner'); ctypes.CDLL(None).fflush(None); print('python chatter'); "
            "bench.emit({'ok': 1})" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == '{"ok": 1}\n' and "banner" in r.stderr and "python chatter" in r.stderr
