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


# ---------------------------------------------------------------------------------------------------------------------------
# The predict loop under torch.distributed, without a GPU: the real `BottomUpPredictor._predict_generator` (two-deep pipeline,
# frame sharding, gather, overflow handling) around a stand-in model whose "network" reads the answer out of the frame.
# Frame t encodes (k_t instances, id t); the stand-in reports SA_STATUS_INSTANCE_OVERFLOW when k_t exceeds the current
# capacity, as the grouping kernel does. Both ranks must double the same caps, re-run, and end with identical, complete results.
def _fake_predictor(max_instances):
    from types import SimpleNamespace

    from sleap_amd import _lib
    from sleap_amd.nn.inference import BottomUpInferenceModel, BottomUpPredictor

    N = 3
    layer = SimpleNamespace(paf_scorer=SimpleNamespace(max_instances=max_instances, n_nodes=N, max_node_peaks=8), max_peaks=64,
                            keras_model=SimpleNamespace(device=torch.device("cpu")), last_upload_done=None)

    class Model(BottomUpInferenceModel):
        def __init__(self):
            self.bottomup_layer = layer
            self.calls = 0

        def call(self, batch):
            self.calls += 1
            x = batch if isinstance(batch, torch.Tensor) else torch.from_numpy(np.asarray(batch))
            b, I = x.shape[0], layer.paf_scorer.max_instances
            k, t = x[:, 0, 0, 0].to(torch.int64), x[:, 0, 1, 0].to(torch.float32)
            peaks = torch.full((b, I, N, 2), float("nan"))
            vals = torch.full((b, I, N), float("nan"))
            scores = torch.full((b, I), float("nan"))
            status = torch.zeros((b,), dtype=torch.int32)
            for f in range(b):
                n = min(int(k[f]), I)
                if int(k[f]) > I:
                    status[f] = _lib.STATUS_INSTANCE_OVERFLOW
                for i in range(n):
                    peaks[f, i] = t[f] + i
                    vals[f, i] = 0.5
                    scores[f, i] = 1.0 + i
            return {"instance_peaks": peaks, "instance_peak_vals": vals, "instance_scores": scores,
                    "n_valid": torch.minimum(k, torch.tensor(I)).to(torch.int32), "status": status}

    pred = BottomUpPredictor.__new__(BottomUpPredictor)
    pred.inference_model, pred.batch_size, pred.verbosity, pred.report_rate, pred.tracker = Model(), 4, "none", 2.0, None
    return pred, layer


def _predict_worker(rank, world, port, q, batch_size=4, T=13):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # T = 13, batch 4: 4 batches, the last one short (1 frame: rank 1 gets an empty shard)
        ks = ([1, 2, 0, 2, 3, 1, 5, 2, 0, 1, 4, 2, 5] * 3)[:T]
        frames = np.zeros((T, 4, 4, 1), np.uint8)
        frames[:, 0, 0, 0], frames[:, 0, 1, 0] = ks, np.arange(T)
        pred, layer = _fake_predictor(max_instances=2)
        pred.batch_size = batch_size
        exs = list(pred._predict_generator(frames))
        assert [len(e["frame_ind"]) for e in exs] == [min(batch_size, T - i) for i in range(0, T, batch_size)]
        n_valid = np.concatenate([e["n_valid"] for e in exs])
        frame_ind = np.concatenate([e["frame_ind"] for e in exs])
        ok = n_valid.tolist() == ks and frame_ind.tolist() == list(range(T))
        for e in exs:
            for f, t in enumerate(e["frame_ind"]):
                for i in range(int(e["n_valid"][f])):
                    ok = ok and bool((e["instance_peaks"][f, i] == t + i).all()) and e["instance_scores"][f, i] == 1.0 + i
        q.put((rank, ok, layer.paf_scorer.max_instances, pred.inference_model.calls))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_predict_loop_overflow_consensus(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_predict_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert all(ok for _, ok, _, _ in res), res
    assert [caps for _, _, caps, _ in res] == [8] * world, res  # 2 -> 4 (k = 3) -> 8 (k = 5): the same on every rank


def test_predict_loop_world_8_ragged_batches_and_empty_shards():
    """The 8-GPU layout of BASELINE configs[3] on CPU: 8 gloo ranks, a global batch that is NOT a multiple of 8 (11 frames:
    ceil-sized shards of 2, 2, 2, 2, 2, 1, 0, 0) and a short last batch of 5 (1, 1, 1, 1, 1, 0, 0, 0) -- every rank ends with the
    complete, ordered result and the same grown capacities; ranks with empty shards still take part in every gather."""
    world, ctx = 8, mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_predict_worker, args=(r, world, port, q, 11, 27)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert [r for r, _, _, _ in res] == list(range(world))
    assert all(ok for _, ok, _, _ in res), res
    assert [caps for _, _, caps, _ in res] == [8] * world, res
    # shards of rank r in the 11-frame batches: the stand-in model is only called where the shard is non-empty
    assert res[7][3] == 0 and res[6][3] == 0 and res[0][3] >= 3


# The base generator (top-down / single-instance predictors): frame sharding + all_gather_object of ragged NumPy results, with
# batch k+1 queued before batch k is converted. Stand-in model: instance count and coordinates are read out of the frame.
def _base_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sleap_amd.nn.inference import Predictor

        class Model:
            order = []

            def predict_on_batch(self, x, numpy=False):
                x = np.asarray(x)
                b, k = x.shape[0], x[:, 0, 0, 0].astype(np.int64)
                imax = int(k.max())  # ragged across batches AND ranks: the gather pads to the widest
                peaks = torch.full((b, imax, 2, 2), float("nan"))
                for f in range(b):
                    peaks[f, : k[f]] = float(x[f, 0, 1, 0])
                self.order.append(("submit", int(x[0, 0, 1, 0])))
                return {"instance_peaks": peaks, "instance_peak_vals": peaks[..., 0].clone(), "n_valid": torch.from_numpy(k)}

            def outputs_to_numpy(self, outs):
                self.order.append(("convert", int(np.nanmin(outs["instance_peaks"].numpy()[0])) if outs["n_valid"][0] > 0 else -1))
                return {k: v.numpy() for k, v in outs.items()}

        T = 11
        ks = [1, 2, 3, 1, 2, 1, 4, 1, 2, 2, 3]
        frames = np.zeros((T, 4, 4, 1), np.uint8)
        frames[:, 0, 0, 0], frames[:, 0, 1, 0] = ks, np.arange(T) + 1
        pred = Predictor.__new__(Predictor)
        pred.inference_model, pred.batch_size = Model(), 4
        exs = list(pred._predict_generator(frames))
        ok = np.concatenate([e["n_valid"] for e in exs]).tolist() == ks
        ok = ok and np.concatenate([e["frame_ind"] for e in exs]).tolist() == list(range(T))
        for e in exs:
            for f, t in enumerate(e["frame_ind"]):
                n = int(e["n_valid"][f])
                ok = ok and bool((e["instance_peaks"][f, :n] == t + 1).all()) and bool(np.isnan(e["instance_peaks"][f, n:]).all())
        # pipelining: the second batch is submitted before the first is converted
        kinds = [a for a, _ in pred.inference_model.order]
        ok = ok and kinds[:3] == ["submit", "submit", "convert"]
        q.put((rank, bool(ok)))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_base_predict_loop_shards_gathers_and_pipelines(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_base_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=10) for _ in range(world)) == [(r, True) for r in range(world)]
