"""The persistent tile schedule of the 3x3 MFMA kernels (csrc/conv3x3.hip: a workgroup walks several tiles, the first chunk of
its next tile is prefetched across the tile boundary) must not change a single bit: every tile's arithmetic is the same as
with one workgroup per tile. `sa_conv3x3_set_grid_limit` forces a handful of workgroups over many tiles -- uneven tails,
workgroup counts that are not multiples of 8 (the XCD split), odd and even chunk counts (the LDS stage parity carries across
tiles), two-source K loops, pooled stores, fused heads -- at sizes far below the point where the automatic policy turns
persistent (occupancy x 256 CUs tiles)."""
import ctypes as C

import numpy as np
import pytest

from parity_helpers import STORAGE_DTYPES  # noqa: E402
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def grid_limit():
    from sleap_amd import _lib

    libs = [_lib.lib("fp16"), _lib.lib("bf16")]

    def set_(n):
        for h in libs:
            h.sa_conv3x3_set_grid_limit(int(n))

    for h in libs:  # the two-workgroup persistent kernels are off by default (slower): these tests keep them correct
        h.sa_conv3x3_set_persistent(1)
    yield set_
    set_(0)
    for h in libs:
        h.sa_conv3x3_set_persistent(-1)


@pytest.mark.parametrize("dtype", STORAGE_DTYPES)
@pytest.mark.parametrize("B,H,W,C0,C1,Cout,mode,pooled", [
    (3, 48, 96, 32, 0, 64, 0, True),     # 2 chunks, 27 tiles, pooled + full
    (2, 40, 70, 64, 0, 64, 0, False),    # 4 chunks, ragged right/bottom tiles
    (2, 32, 64, 48, 0, 128, 0, False),   # 3 chunks (odd: stage parity flips per tile), 2 cout tiles
    (1, 64, 64, 64, 128, 64, 1, False),  # concat: 12 chunks from two sources
    (5, 16, 32, 256, 0, 32, 0, True),    # 16 chunks, one cout tile of 32
])
def test_persistent_schedule_is_bitwise_neutral(grid_limit, dtype, B, H, W, C0, C1, Cout, mode, pooled):
    from sleap_amd import ops

    g = torch.Generator(device="cpu").manual_seed(B * 131 + H + C0 + C1 + Cout)
    k = torch.randn((3, 3, C0 + C1, Cout), generator=g) * (2.0 / (9 * (C0 + C1))) ** 0.5
    x0 = ops.to_bf16_padded(torch.randn((B, H, W, C0), generator=g).cuda(), dtype=dtype)
    x1 = ops.to_bf16_padded(torch.randn((B, H, W, C1), generator=g).cuda(), dtype=dtype) if C1 else None
    pw = ops.pack_conv3x3_weights(k.numpy(), C0, C1, dtype=dtype)
    coutp = ops.pad16(Cout)
    bias = torch.zeros((coutp,), dtype=torch.float32)
    bias[:Cout] = torch.randn((Cout,), generator=g) * 0.1
    bias = bias.cuda()

    def run():
        # caller-owned outputs, poisoned: a tile that a schedule skipped must not look computed (round 5: with fewer than 8
        # workgroups whole XCD ranges of tiles were left out, and the allocator's stale -- correct -- bytes hid it)
        o_full = torch.full((B, H, W, coutp), float("nan"), dtype=x0.dtype, device="cuda")
        o_pool = torch.full((B, H // 2, W // 2, coutp), float("nan"), dtype=x0.dtype, device="cuda") if pooled else None
        out = ops.conv3x3(x0, x1, mode, pw, bias, coutp, True, (H, W), full=True, pooled=pooled, out=o_full, out_pool=o_pool)
        torch.cuda.synchronize()
        outs = [o.clone() for o in (out if pooled else (out,))]
        for o in outs:
            assert not torch.isnan(o.float()).any(), "a tile was not written"
        return outs

    grid_limit(-1)  # one workgroup per tile
    ref = run()
    n_tiles = B * ((H + 15) // 16) * ((W + 31) // 32) * ((coutp + 63) // 64)
    for n in (1, 3, 7, 8, 13, n_tiles - 1):
        if n < 1:
            continue
        grid_limit(n)
        for a, b in zip(run(), ref):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16)), f"grid limit {n} of {n_tiles} tiles"


@pytest.mark.parametrize("B,H,W,C0,C1,Cout,pooled,limit", [
    (8, 256, 256, 64, 0, 64, True, 0),      # the 64 -> 64 @256 layer of the benchmark plan at 8 frames: 1024 tiles on 512 workgroups
    (8, 256, 256, 32, 0, 64, False, 96),    # 2 chunks (the shortest K loop: the wait is reached soonest after the stores)
    (16, 128, 128, 64, 0, 128, False, 0),   # 4 chunks, 2 cout tiles
    (16, 64, 64, 256, 0, 256, False, 200),  # 16 chunks (copies queued mid-chunk), 4 cout tiles, 200 workgroups: not a multiple of 8
    (6, 128, 128, 128, 256, 128, False, 64),  # concat, 24 chunks
])
def test_counted_wait_of_the_two_workgroup_persistent_loop_at_real_sizes(grid_limit, B, H, W, C0, C1, Cout, pooled, limit):
    """Round 5 (`PERS` kernels, csrc/conv3x3.hip): in front of a later tile's first chunk a wave waits with `s_waitcnt vmcnt(S)`
    -- S = the store instructions of the previous tile's epilogue, issued AFTER the next tile's copies -- instead of
    vmcnt(0). That relies on the in-order retirement of gfx9's one VMEM counter; if a copy could still be in flight behind
    that wait, tiles would be computed from a stale LDS stage. Full-size layers, every tile inside the image (the counted path;
    ragged tiles fall back to vmcnt(0)), many tiles per workgroup, three runs each: bitwise the one-workgroup-per-tile result."""
    from sleap_amd import ops

    dtype = "fp16"
    g = torch.Generator(device="cpu").manual_seed(B + H + C0 + C1 + Cout)
    k = torch.randn((3, 3, C0 + C1, Cout), generator=g) * (2.0 / (9 * (C0 + C1))) ** 0.5
    x0 = ops.to_bf16_padded(torch.randn((B, H, W, C0), generator=g).cuda(), dtype=dtype)
    x1 = ops.to_bf16_padded(torch.randn((B, H, W, C1), generator=g).cuda(), dtype=dtype) if C1 else None
    pw = ops.pack_conv3x3_weights(k.numpy(), C0, C1, dtype=dtype)
    coutp = ops.pad16(Cout)
    bias = (torch.randn((coutp,), generator=g) * 0.1).cuda()
    from sleap_amd import _lib

    mode = (1 if C1 else 0) | _lib.LAYOUT_PLANES16  # the layout of the benchmark plan (the same bytes read as 16-channel planes)

    def run():
        o_full = torch.full((B, H, W, coutp), float("nan"), dtype=x0.dtype, device="cuda")  # poisoned: a skipped tile shows
        o_pool = torch.full((B, H // 2, W // 2, coutp), float("nan"), dtype=x0.dtype, device="cuda") if pooled else None
        out = ops.conv3x3(x0, x1, mode, pw, bias, coutp, True, (H, W), full=True, pooled=pooled, out=o_full, out_pool=o_pool)
        torch.cuda.synchronize()
        return [o.clone() for o in (out if pooled else (out,))]

    grid_limit(-1)  # one workgroup per tile: no tile boundary inside a workgroup, the counted wait is never taken
    ref = run()
    grid_limit(limit)  # 0: the automatic policy (occupancy x CUs workgroups)
    for rep in range(3):
        for a, b in zip(run(), ref):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16)), f"run {rep}"


@pytest.mark.parametrize("dtype", STORAGE_DTYPES)
def test_persistent_network_forward_is_bitwise_neutral(grid_limit, dtype):
    """The benchmark UNet (fused stem / pair / heads included) at a small size: heads identical for every schedule."""
    from sleap_amd.nn.architectures import build_unet_model_config, he_normal_weights
    from sleap_amd.nn.engine import DeviceNetwork
    from sleap_amd.synth import render_frames

    cfg, shapes = build_unet_model_config((160, 224, 1), 16, 2, 32, 4, True, True,
                                          heads=[("MultiInstanceConfmapsHead", 13, 4), ("PartAffinityFieldsHead", 24, 8)])
    net = DeviceNetwork(cfg, he_normal_weights(shapes, seed=3), dtype=dtype)
    x = torch.from_numpy(render_frames(3, 160, 224, n_animals=2, seed=5)[0]).cuda()
    grid_limit(-1)
    ref = [o.clone() for o in net.forward(x)]
    for n in (2, 5, 16):
        grid_limit(n)
        for a, b in zip(net.forward(x), ref):
            assert torch.equal(a, b), f"grid limit {n}"
    grid_limit(0)
    for a, b in zip(net.forward(x), ref):
        assert torch.equal(a, b)
