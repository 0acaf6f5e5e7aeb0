"""Host logic of the range-safe fp16 mode (sleap_amd/nn/range_scaling.py), checked on the CPU against the fp32 oracle graph:
the folded weights compute 2^k(layer) x the original layer outputs, the heads come out unchanged, and the planned exponents
bring every scaled tensor under the limit."""
import numpy as np
import pytest

from oracle.keras_graph import KerasGraph, ensure_float
from sleap_amd.nn import architectures as A
from sleap_amd.nn import range_scaling as R


def _ranges(all_t):
    return {k: float(np.abs(v).max()) for k, v in all_t.items()}


def _check(mc, w, x, expect_scaled, aliases=None, trigger=R.TRIGGER, limit=R.LIMIT):
    outs, all_t = KerasGraph(mc, w)(x, return_all=True)
    rng = _ranges(all_t)
    ks = R.plan_scales(mc, rng, aliases, limit=limit, trigger=trigger)
    assert (max(abs(k) for k in ks.values()) > 0) == expect_scaled
    w2 = R.fold_scales(mc, w, ks)
    assert set(w2) == set(w)
    outs2, all2 = KerasGraph(mc, w2)(x, return_all=True)
    for a, b in zip(outs, outs2):
        assert float(np.abs(a - b).max()) <= 2e-5 * float(np.abs(a).max())
    worst = 0.0
    cats = {l["name"] for l in mc["config"]["layers"] if l["class_name"] == "Concatenate"}
    for name, k in ks.items():
        if name not in all_t or name in cats:  # (a Concatenate output carries one scale per segment)
            continue
        ref = all_t[name] * np.float32(2.0 ** k)
        assert float(np.abs(all2[name] - ref).max()) <= 2e-5 * max(float(np.abs(ref).max()), 1e-30), name
        if k:
            worst = max(worst, float(np.abs(all2[name]).max()))
            assert rng[name] * 2.0 ** k <= limit * 1.0001
    return ks, rng, worst


def test_plain_he_resnet50_is_brought_into_range():
    """Plain He-normal ResNet-50 + transposed-conv upsampling stack with concatenated skips (configs[4]'s architecture,
    BatchNormalization at its initial statistics): activations pass fp16's 65504 at depth; the folded network keeps every
    scaled tensor under the limit and the heads bit-comparable."""
    mc, shapes = A.build_resnet_model_config((96, 96, 1), "ResNet50", 32, True,
                                             upsampling=dict(output_stride=4, method="transposed_conv",
                                                             skip_connections="concatenate"),
                                             heads=[("MultiInstanceConfmapsHead", 5, 4), ("PartAffinityFieldsHead", 8, 8)])
    w = A.he_normal_weights(shapes, seed=3)
    x = ensure_float(np.random.default_rng(0).integers(0, 256, (1, 96, 96, 1), dtype=np.uint8))
    ks, rng, worst = _check(mc, w, x, expect_scaled=True)
    assert max(rng.values()) > 65504.0  # the reason the mode exists
    assert worst <= R.LIMIT * 1.0001
    # model inputs and outputs keep scale 1
    cfg = mc["config"]
    for n in [cfg["input_layers"][0][0]] + [o[0] for o in cfg["output_layers"]]:
        assert ks[n] == 0


def test_network_inside_the_range_is_left_alone():
    mc, shapes = A.build_unet_model_config((64, 64, 1), 8, 2.0, 16, 2, True, True, None,
                                           heads=[("SingleInstanceConfmapsHead", 3, 2)])
    w = A.he_normal_weights(shapes, seed=1)
    x = ensure_float(np.random.default_rng(1).integers(0, 256, (1, 64, 64, 1), dtype=np.uint8))
    outs, all_t = KerasGraph(mc, w)(x, return_all=True)
    ks = R.plan_scales(mc, _ranges(all_t))
    assert set(ks.values()) == {0}
    w2 = R.fold_scales(mc, w, ks)
    assert all(np.array_equal(w[k], w2[k]) for k in w)


def test_unet_concatenate_segments_with_different_scales():
    """A UNet whose decoder concatenates a skip and an upsampled tensor that got DIFFERENT exponents (low trigger): the
    consuming conv scales its kernel rows per segment."""
    mc, shapes = A.build_unet_model_config((64, 64, 1), 8, 2.0, 16, 2, True, True, None,
                                           heads=[("SingleInstanceConfmapsHead", 3, 2)])
    w = A.he_normal_weights(shapes, seed=2)
    for k in w:  # inflate the activations with depth
        if k.endswith("/kernel") and "head" not in k.lower():
            w[k] = w[k] * np.float32(2.5)
    x = ensure_float(np.random.default_rng(2).integers(0, 256, (1, 64, 64, 1), dtype=np.uint8))
    ks, rng, _ = _check(mc, w, x, expect_scaled=True, trigger=4.0, limit=1.0)
    cats = [l for l in mc["config"]["layers"] if l["class_name"] == "Concatenate"]
    assert cats and any(len({ks[n[0]] for n in l["inbound_nodes"][0]}) > 1 for l in cats)


def test_hourglass_add_operands_share_a_scale():
    mc, shapes = A.build_hourglass_model_config((64, 64, 1), stem_stride=4, max_stride=32, output_stride=4, stem_filters=16, filters=16,
                                                filter_increase=8, stacks=1,
                                                heads=[("SingleInstanceConfmapsHead", 3, 4)])
    w = A.he_normal_weights(shapes, seed=4)
    for k in w:
        if k.endswith("/kernel"):
            w[k] = w[k] * np.float32(1.6)
    x = ensure_float(np.random.default_rng(3).integers(0, 256, (1, 64, 64, 1), dtype=np.uint8))
    ks, _, _ = _check(mc, w, x, expect_scaled=True, trigger=8.0, limit=2.0)
    for l in mc["config"]["layers"]:
        if l["class_name"] == "Add":
            assert len({ks[n[0]] for n in l["inbound_nodes"][0]} | {ks[l["name"]]}) == 1


# ---- ranks of a frame-sharded run must fold the SAME exponents: the ranges go through a MAX all-reduce
def _dist_worker(rank, world, port, q):
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0's shard stays small, rank 1's overflows one tensor; a NaN on one rank must win as inf
        mine = [100.0, 3.0e5, 7.0] if rank else [900.0, 2.0, float("nan")]
        red = R.dist_max(mine)
        mc, all_t = _toy_resnet()
        rng = _ranges(all_t)
        local = {k: (v * (50.0 if rank else 1.0)) for k, v in rng.items()}
        keys = sorted(local)
        merged = dict(zip(keys, R.dist_max([local[k] for k in keys])))
        q.put((rank, red, R.plan_scales(mc, merged)))
    finally:
        dist.destroy_process_group()


def _toy_resnet():
    from oracle.keras_graph import KerasGraph
    from sleap_amd.nn.architectures import build_resnet_model_config, he_normal_weights

    mc, shapes = build_resnet_model_config((64, 64, 1), "ResNet50", 32, True,
                                           upsampling=dict(output_stride=4, method="transposed_conv", skip_connections="concatenate"),
                                           heads=[("MultiInstanceConfmapsHead", 3, 4)])
    w = he_normal_weights(shapes, seed=1)
    x = np.random.default_rng(0).random((1, 64, 64, 1)).astype(np.float32)
    return mc, KerasGraph(mc, w)(x, return_all=True)[1]


def test_ranks_agree_on_ranges_and_exponents_gloo_world_2():
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=240) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(60)
    (_, red0, ks0), (_, red1, ks1) = got
    assert red0 == red1 == [900.0, 3.0e5, float("inf")]
    assert ks0 == ks1 and min(ks0.values()) < 0  # rank 1's larger activations decide for both


# ---- the agreement point of a frame-sharded run: one decision for all ranks, an empty shard takes part with neutral values
def _agree_worker(rank, world, port, case, q):
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mc, all_t = _toy_resnet()
        base = _ranges(all_t)
        names = sorted(base)
        calls = {"measure": 0}

        def measure():
            calls["measure"] += 1
            return {k: v * (60.0 if rank == 1 else 1.0) for k, v in base.items()}

        if case == "fits":  # nobody near the range: one collective, nothing measured, nothing installed
            scan = (900.0 + rank, False)
            has = True
        elif case == "rank1_overflows_rank2_empty":  # rank 1 saw inf, rank 2 has no frames at all
            scan = {0: (500.0, False), 1: (60000.0, True), 2: None}[rank]
            has = rank != 2
        else:  # "already_scaled": an overflow with explicit exponents in place is an error on EVERY rank
            scan = (100.0, rank == 0)
            has = True
        try:
            ks = R.dist_agree(scan if has else None, case == "already_scaled", True, lambda: names, measure if has else None,
                              lambda r: R.plan_scales(mc, r))
            q.put((rank, "ok", ks, calls["measure"]))
        except FloatingPointError:
            q.put((rank, "raised", None, calls["measure"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("fits", 2), ("rank1_overflows_rank2_empty", 3), ("already_scaled", 2)])
def test_ranks_agree_once_at_one_program_point_gloo(case, world):
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_agree_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=240) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(60)
    if case == "fits":
        assert all(g[1] == "ok" and g[2] is None and g[3] == 0 for g in got)
    elif case == "rank1_overflows_rank2_empty":
        assert all(g[1] == "ok" for g in got)
        assert got[0][2] == got[1][2] == got[2][2] and min(got[0][2].values()) < 0  # the same exponents everywhere
        assert [g[3] for g in got] == [1, 1, 0]  # ranks with frames measured, the empty one did not
    else:
        assert all(g[1] == "raised" for g in got)
