"""N>1 path on CPU: world_size-2 gloo processes shard a batch by contiguous frame ranges and assemble the
packed results with one all-gather (sleap_amd/parallel.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sleap_amd import parallel


def test_shard_range_covers_batch_contiguously():
    for n in (1, 2, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = parallel.shard_range(100, 100 + n, r, world)
                got.extend(range(lo, hi))
            assert got == list(range(100, 100 + n))


def test_pack_unpack_roundtrip():
    I, N, b = 5, 13, 3
    rng = np.random.default_rng(0)
    outs = {"instance_peaks": torch.from_numpy(rng.random((b, I, N, 2)).astype(np.float32)),
            "instance_peak_vals": torch.from_numpy(rng.random((b, I, N)).astype(np.float32)),
            "instance_scores": torch.from_numpy(rng.random((b, I)).astype(np.float32)),
            "n_valid": torch.tensor([0, 3, 5], dtype=torch.int32), "status": torch.tensor([0, 16, 0], dtype=torch.int32)}
    outs["instance_peaks"][0] = float("nan")
    p = parallel.pack_results(outs)
    assert p.shape == (b, parallel.packed_width(I, N))
    u = parallel.unpack_results(p, I, N)
    for k in outs:
        np.testing.assert_array_equal(u[k].numpy(), outs[k].numpy())
    # the NumPy twin used on the host side of the predict loop (no torch CPU ops there: container CPU quotas)
    un = parallel.unpack_results_np(p.numpy(), I, N)
    assert set(un) == set(u)
    for k in outs:
        np.testing.assert_array_equal(un[k], outs[k].numpy())
        assert un[k].dtype == outs[k].numpy().dtype


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_batch, I, N, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        width = parallel.packed_width(I, N)
        full = torch.arange(n_batch * width, dtype=torch.float32).reshape(n_batch, width)
        assert parallel.rank_world() == (rank, world)
        lo, hi = parallel.shard_range(0, n_batch, rank, world)
        mine = full[lo:hi].clone() if hi > lo else None
        got = parallel.gather_batch_results(mine, n_batch, I, N, world)
        q.put((rank, bool(torch.equal(got, full))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_batch", [8, 7, 1])
def test_two_rank_gather(n_batch):
    world, I, N = 2, 4, 13
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_batch, I, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(world))
    assert res == {0: True, 1: True}
