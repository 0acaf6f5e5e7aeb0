"""Network-layer kernels vs a plain torch fp32 reference of the same op (on the same bf16-rounded
operands), and whole-graph parity vs the fp32 CPU oracle (oracle/keras_graph.py).

Tolerances: a bf16 store has 8 mantissa bits (rel. 2^-9 rounding error); layer tests compare against
the fp32 result of identical bf16 inputs, so only accumulation order and the final rounding differ:
|delta| <= 1e-2 * max|ref| is generous. Whole-network tolerance is stated in the test.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# the suite follows the storage type of the library build under test (SLEAP_AMD_DTYPE, default fp16); the tolerances below were
# written for bf16 (8 mantissa bits) and hold a fortiori for fp16 (11)
from parity_helpers import STORAGE_DTYPES  # noqa: E402
from sleap_amd import _lib as _L  # noqa: E402

TD = {"bf16": torch.bfloat16, "fp16": torch.float16}[_L.DEFAULT_DTYPE]

MODELS = os.path.join(os.path.dirname(__file__), "golden", "models")


def _bf(x):
    return x.to(TD).to(torch.float32)


def _ref_conv(x_nhwc, k_keras, bias, relu):
    y = F.conv2d(x_nhwc.permute(0, 3, 1, 2), k_keras.permute(3, 2, 0, 1), bias, padding=1)
    if relu:
        y = torch.relu(y)
    return y.permute(0, 2, 3, 1)


CONV_CASES = [
    # (B, H, W, C0, C1, Cout, mode)  mode: 0 none, 1 concat, 2 concat+upsample, 4 pool
    (2, 16, 32, 16, 0, 16, 0),
    (1, 37, 45, 32, 0, 64, 0),  # ragged tiles
    (1, 20, 70, 24, 0, 36, 0),  # channel padding (24->32, 36->48), CK=16 path
    (2, 32, 32, 64, 0, 128, 0),  # two cout tiles per WG + cout grid
    (1, 16, 16, 256, 0, 96, 0),
    (1, 24, 40, 32, 32, 32, 1),
    (1, 24, 40, 36, 54, 36, 1),
    (2, 32, 64, 64, 128, 64, 2),
    (1, 16, 32, 24, 36, 24, 2),
    (1, 32, 32, 16, 0, 32, 4),
    (2, 18, 34, 48, 0, 16, 4),
]


@pytest.mark.parametrize("B,H,W,C0,C1,Cout,mode", CONV_CASES)
def test_conv3x3_vs_torch(B, H, W, C0, C1, Cout, mode):
    from sleap_amd import ops

    g = torch.Generator(device="cpu").manual_seed(B * 1000 + H * 10 + C0 + C1 + Cout + mode)
    k = torch.randn((3, 3, C0 + C1, Cout), generator=g) * (2.0 / (9 * (C0 + C1))) ** 0.5
    bias = torch.randn((Cout,), generator=g) * 0.1
    if mode == 4:
        x0 = torch.randn((B, 2 * H, 2 * W, C0), generator=g)
    else:
        x0 = torch.randn((B, H, W, C0), generator=g)
    x1 = None
    if mode == 1:
        x1 = torch.randn((B, H, W, C1), generator=g)
    elif mode == 2:
        x1 = torch.randn((B, H // 2, W // 2, C1), generator=g)
    # reference on bf16-rounded operands
    r0 = _bf(x0)
    if mode == 4:
        r0 = F.max_pool2d(r0.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    rin = r0
    if x1 is not None:
        r1 = _bf(x1)
        if mode == 2:
            r1 = _bf(F.interpolate(r1.permute(0, 3, 1, 2), scale_factor=2.0, mode="bilinear", align_corners=False)
                     .permute(0, 2, 3, 1))
        rin = torch.cat([r0, r1], dim=-1)
    ref = _ref_conv(rin, _bf(k), bias, True)
    # device
    d0 = ops.to_bf16_padded(x0.cuda().contiguous())
    d1 = ops.to_bf16_padded(x1.cuda().contiguous()) if x1 is not None else None
    pw = ops.pack_conv3x3_weights(k.numpy(), C0, C1)
    coutp = ops.pad16(Cout)
    bp = torch.zeros((coutp,), dtype=torch.float32)
    bp[:Cout] = bias
    pooled = mode in (0, 1) and H % 2 == 0 and W % 2 == 0
    out = ops.conv3x3(d0, d1, mode, pw, bp.cuda(), coutp, True, (H, W), full=True, pooled=pooled)
    if pooled:
        out, outp = out
        # fused MaxPool2D(2) epilogue == pooling the stored full-resolution output, exactly
        want = F.max_pool2d(out.float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
        assert torch.equal(outp.float(), want)
        only = ops.conv3x3(d0, d1, mode, pw, bp.cuda(), coutp, True, (H, W), full=False, pooled=True)
        assert torch.equal(only.float(), want)
    got = ops.from_bf16(out, Cout).cpu()
    pad = out.float()[..., Cout:]
    assert float(pad.abs().max()) == 0.0 if pad.numel() else True
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 1e-2 * scale, (float((got - ref).abs().max()), scale)


def test_conv3x3_identity_asymmetric():
    """Delta kernel with an asymmetric tap: catches transposed MFMA operand/accumulator layouts."""
    from sleap_amd import ops

    C = 32
    k = torch.zeros((3, 3, C, C))
    for c in range(C):
        k[0, 2, c, (c + 5) % C] = 1.0  # out[y,x,(c+5)%C] = in[y-1,x+1,c]
    x = torch.randn((1, 8, 40, C))
    d0 = ops.to_bf16_padded(x.cuda())
    out = ops.conv3x3(d0, None, 0, ops.pack_conv3x3_weights(k.numpy(), C), torch.zeros(C).cuda(), C, False, (8, 40))
    got = ops.from_bf16(out, C).cpu()
    ref = torch.zeros_like(x)
    xs = _bf(x)
    ref[:, 1:, :-1, :] = xs[:, :-1, 1:, :]
    ref = ref[..., [(c - 5) % C for c in range(C)]]
    assert torch.equal(got, ref)


def _close(got, ref, rel, abs_=0.0):
    err = float((got - ref).abs().max())
    lim = rel * float(ref.abs().max()) + abs_
    assert err <= lim, f"max|delta|={err:.4g} > {lim:.4g}"


def test_small_layers_vs_torch():
    from sleap_amd import _lib, ops
    from sleap_amd.ops import _ptr, _stream, check

    h = _lib.lib()
    g = torch.Generator().manual_seed(0)
    # stem u8 (device tensors are kept alive in named variables until the result is read back)
    img = torch.randint(0, 256, (2, 20, 36, 1), generator=g, dtype=torch.uint8)
    k = torch.randn((3, 3, 1, 16), generator=g)
    b = torch.randn((16,), generator=g)
    d_img, d_k, d_b = img.cuda(), k.cuda().contiguous(), b.cuda()
    out = torch.empty((2, 20, 36, 16), dtype=TD, device="cuda")
    check(h.sa_stem_conv3x3(_ptr(d_img), 1, 2, 20, 36, 1, _ptr(d_k), _ptr(d_b), 16, 1, _ptr(out), _stream()), "stem")
    _close(out.float().cpu(), _ref_conv(img.float() * np.float32(1 / 255), k, b, True), 1e-2)
    # stem f32 rgb
    imgf = torch.rand((1, 12, 12, 3), generator=g)
    k3 = torch.randn((3, 3, 3, 8), generator=g)
    b3 = torch.randn((8,), generator=g)
    d_img, d_k, d_b = imgf.cuda(), k3.cuda().contiguous(), b3.cuda()
    out = torch.empty((1, 12, 12, 8), dtype=TD, device="cuda")
    check(h.sa_stem_conv3x3(_ptr(d_img), 0, 1, 12, 12, 3, _ptr(d_k), _ptr(d_b), 8, 0, _ptr(out), _stream()), "stem")
    _close(out.float().cpu(), _ref_conv(imgf, k3, b3, False), 1e-2)
    # pool / upsample
    x = torch.randn((2, 12, 20, 32), generator=g)
    xd = ops.to_bf16_padded(x.cuda())
    o = torch.empty((2, 6, 10, 32), dtype=TD, device="cuda")
    check(h.sa_maxpool2x2_bf16(_ptr(xd), 2, 12, 20, 32, _ptr(o), _stream()), "pool")
    ref = F.max_pool2d(_bf(x).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    assert torch.equal(o.float().cpu(), ref)
    for bil, modeN in ((1, "bilinear"), (0, "nearest")):
        o = torch.empty((2, 24, 40, 32), dtype=TD, device="cuda")
        check(h.sa_upsample2x_bf16(_ptr(xd), 2, 12, 20, 32, bil, _ptr(o), _stream()), "up")
        kw = dict(align_corners=False) if bil else {}
        ref = F.interpolate(_bf(x).permute(0, 3, 1, 2), scale_factor=2.0, mode=modeN, **kw).permute(0, 2, 3, 1)
        _close(o.float().cpu(), ref, 0.0, 2e-2)
    # 1x1 head
    wk = torch.randn((13, 32), generator=g)
    bk = torch.randn((13,), generator=g)
    d_w, d_b = wk.cuda(), bk.cuda()
    o = torch.empty((2, 12, 20, 13), dtype=torch.float32, device="cuda")
    check(h.sa_conv1x1_head(_ptr(xd), 32, _ptr(d_w), _ptr(d_b), 13, 0, 2, 12, 20, _ptr(o), _stream()), "head")
    _close(o.cpu(), _bf(x) @ wk.T + bk, 1e-4, 1e-5)
    # ... with 33-64 output channels (the 46-channel PAF head of a 23-edge skeleton, BASELINE configs[4]): two 32-channel tiles
    # of the matrix-core kernel; > 64 channels: the VALU kernel
    for cout in (46, 64, 70):
        wk2 = torch.randn((cout, 32), generator=g)
        bk2 = torch.randn((cout,), generator=g)
        o2 = torch.empty((2, 12, 20, cout), dtype=torch.float32, device="cuda")
        d_w2, d_b2 = wk2.cuda(), bk2.cuda()  # (kept alive: a temporary's memory would be handed to the next allocation)
        check(h.sa_conv1x1_head(_ptr(xd), 32, _ptr(d_w2), _ptr(d_b2), cout, 0, 2, 12, 20, _ptr(o2), _stream()), "head")
        _close(o2.cpu(), _bf(x) @ wk2.T + bk2, 1e-4, 1e-5)
    # transposed conv k3 s2 same == full transposed conv cropped at the end
    kt = torch.randn((3, 3, 16, 32), generator=g) * 0.1  # (kh, kw, Cout, Cin)
    bt = torch.randn((16,), generator=g)
    d_w, d_b = kt.cuda().to(TD).contiguous(), bt.cuda()
    o = torch.empty((2, 24, 40, 16), dtype=TD, device="cuda")
    check(h.sa_convt3x3s2_bf16(_ptr(xd), 32, _ptr(d_w), _ptr(d_b), 16, 1, 2, 12, 20, _ptr(o), _stream()), "convt")
    ref = F.conv_transpose2d(_bf(x).permute(0, 3, 1, 2), _bf(kt).permute(3, 2, 0, 1), None, stride=2)[:, :, :24, :40]
    ref = torch.relu(ref + bt.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    _close(o.float().cpu(), ref, 1e-2)


def _run_graph_parity(cfg, weights, x_u8, tol, skip=()):
    """Device heads vs the fp32 oracle and vs the oracle that emulates bf16 storage at the engine's rounding
    points. Error metric: max|delta| / max|ref| per head. (Even the bf16-emulating oracle can not be matched
    tightly at the heads: fp32 summation-order differences flip bf16 roundings by one ulp in a few elements per
    layer and the flips cascade -- see test_layerwise_vs_bf16_emulating_oracle for the per-layer picture.)"""
    from oracle.keras_graph import KerasGraph, ensure_float
    from sleap_amd.nn.engine import DeviceNetwork

    net = DeviceNetwork(cfg, weights)
    outs = [o.cpu().numpy() for o in net.forward(torch.from_numpy(x_u8).cuda())]
    xin = ensure_float(x_u8)
    worst = {}
    for mode in ("bf16", "fp32"):
        ref = KerasGraph(cfg, weights, emulate_bf16=(mode == "bf16"))(xin)
        assert len(outs) == len(ref)
        w = 0.0
        for name, o, r in zip(net.output_names, outs, ref):
            assert o.shape == r.shape
            if any(sk in name for sk in skip):
                continue
            assert np.isfinite(o).all()  # max() below would silently drop a NaN
            w = max(w, float(np.abs(o - r).max() / max(np.abs(r).max(), 1e-6)))
        worst[mode] = w
        assert w <= tol, f"{mode}: max rel err {w:.4g} > {tol}"
    return worst


def _fly_frames(n, h, w, seed):
    from sleap_amd.synth import render_frames

    return render_frames(n, h, w, n_animals=2, seed=seed)[0]


def test_fixture_bottomup_network_vs_oracle():
    """Trained fixture (Conv2DTranspose decoder, 3 heads) on fly-like frames: max|delta|/max|ref| <= 3e-2."""
    from sleap_amd.nn.engine import load_keras_npz

    cfg, w = load_keras_npz(os.path.join(MODELS, "minimal_instance.UNet.bottomup", "best_model.npz"))
    _run_graph_parity(cfg, w, _fly_frames(2, 192, 256, 0), 5e-2)


def test_fixture_bilinear_network_vs_oracle():
    """Trained fixture with the bilinear-upsampling decoder. Its features reach |x| ~ 100 on these (out of
    distribution) frames, so one bf16 ulp of a feature is 0.5 and the linear head, a sum with heavy cancellation,
    shows ~4 % of its range; every bf16 layer itself is within 2 ulp (tests/diagnostics/layer_diff.py). The sigmoid
    ClassMapsHead (identity head, out of scope) is executed but not compared."""
    from sleap_amd.nn.engine import load_keras_npz

    cfg, w = load_keras_npz(os.path.join(MODELS, "min_tracks_2node.UNet.bottomup_multiclass", "best_model.npz"))
    _run_graph_parity(cfg, w, _fly_frames(1, 128, 192, 1), 6e-2, skip=("ClassMapsHead",))


def _benchmark_unet(h, w):
    from sleap_amd.nn.architectures import build_unet_model_config, he_normal_weights

    cfg, shapes = build_unet_model_config((h, w, 1), 16, 2, 32, 4, True, True,
                                          heads=[("MultiInstanceConfmapsHead", 13, 4), ("PartAffinityFieldsHead", 24, 8)])
    return cfg, he_normal_weights(shapes, seed=1)


def test_random_benchmark_unet_vs_oracle():
    """baseline_medium_rf.bottomup topology (f16 r2 s32->4, bilinear) with He-normal weights at 128x160."""
    cfg, w = _benchmark_unet(128, 160)
    rng = np.random.default_rng(2)
    _run_graph_parity(cfg, w, rng.integers(0, 256, (2, 128, 160, 1), dtype=np.uint8), 3e-2)


def test_layerwise_vs_bf16_emulating_oracle():
    """Every bf16 activation tensor of the engine vs the oracle rounding at the same points: the first layers
    differ in a handful of elements by exactly one bf16 ulp; no layer is off by more than 2 ulp of its range."""
    from oracle.keras_graph import KerasGraph, ensure_float
    from sleap_amd.nn.engine import DeviceNetwork

    cfg, w = _benchmark_unet(128, 160)
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, (2, 128, 160, 1), dtype=np.uint8)
    net = DeviceNetwork(cfg, w)
    net.forward(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    assert net.planar  # the UNet family runs on 16-channel planes; stored_tensor() hands back NHWC views / copies
    _, ref = KerasGraph(cfg, w, emulate_bf16=True)(ensure_float(x), return_all=True)
    conv_layers = [l["name"] for l in cfg["config"]["layers"] if l["class_name"] == "Conv2D"]
    acts = {l["inbound_nodes"][0][0][0]: l["name"] for l in cfg["config"]["layers"] if l["class_name"] == "Activation"}
    i = 0
    checked = 0
    for op in net.plan:
        if op[0] in ("stem2", "pair"):  # two convs in one launch: the first one's activation only ever lives in LDS
            i += 1
            op = op[2]
        if op[0] not in ("stem", "conv"):
            continue
        name = conv_layers[i]
        i += 1
        checked += 1
        o = op[1] if op[0] == "stem" else op[6]
        r = ref[acts.get(name, name)]
        if o.buf is None and op[8] is None:
            continue  # consumed only by fused heads: never stored
        if o.buf is None:  # only the fused max-pooled copy of this layer is stored
            o = op[8]
            r = r.reshape(r.shape[0], r.shape[1] // 2, 2, r.shape[2] // 2, 2, r.shape[3]).max(axis=(2, 4))
        d = net.stored_tensor(o.buf, (2, 128, 160)).float().cpu().numpy()[..., : o.c]
        scale = np.abs(r).max()
        err = np.abs(d - r).max() / scale
        assert err <= 2.0 ** -6, (name, err)
        if i <= 2:  # the first two conv LAYERS (not launches: fused launches skip layers whose output never leaves LDS)
            frac = (np.abs(d - r) > 1e-6 * scale).mean()
            assert frac < 2e-3 and err <= 2.0 ** -7 * 1.01, (name, frac, err)


@pytest.mark.parametrize("fuse", [dict(fuse_stem=False), dict(use_stem16=False), dict(fuse_heads=False), dict(fuse_upsample=True),
                                  dict(fuse_upsample=True, layout="nhwc"), dict(fuse_stem=False, fuse_heads=False)])
def test_fusion_variants_agree(fuse):
    """Every fusion switch of the engine computes the same network up to bf16 rounding flips: the fused stem
    evaluates the first conv on the matrix cores (3-term bf16 split of the fp32 weights, fp32-accurate), fused
    heads see un-rounded features."""
    from sleap_amd.nn.engine import DeviceNetwork

    cfg, w = _benchmark_unet(96, 128)
    x = torch.from_numpy(_fly_frames(2, 96, 128, 5)).cuda()
    base = [o.clone() for o in DeviceNetwork(cfg, w).forward(x)]
    other = [o.clone() for o in DeviceNetwork(cfg, w, **fuse).forward(x)]
    for a, b in zip(base, other):
        if False:
            pass
        else:
            assert float((a - b).abs().max()) <= 2e-2 * float(a.abs().max())


@pytest.mark.parametrize("layout", [None, "nhwc"])
def test_fused_heads_with_more_than_32_channels(layout):
    """Round 4: a fused head may have up to 64 channels (two 32-row passes of the matrix-core head GEMM): 24 nodes / 23 edges
    = 24 + 46 channels (BASELINE configs[4]'s heads on a UNet). Fused == stand-alone head launches up to the head GEMM's
    rounding, both within the usual tolerance of the fp32 oracle."""
    from oracle.keras_graph import KerasGraph, ensure_float
    from sleap_amd.nn.architectures import build_unet_model_config, he_normal_weights
    from sleap_amd.nn.engine import DeviceNetwork

    cfg, shapes = build_unet_model_config((96, 128, 1), 16, 2, 32, 4, True, True,
                                          heads=[("MultiInstanceConfmapsHead", 40, 4), ("PartAffinityFieldsHead", 46, 8)])
    w = he_normal_weights(shapes, seed=3)
    x = torch.from_numpy(_fly_frames(2, 96, 128, 9)).cuda()
    ref = KerasGraph(cfg, w)(ensure_float(x.cpu().numpy()))
    fused = DeviceNetwork(cfg, w, layout=layout)
    plain = DeviceNetwork(cfg, w, layout=layout, fuse_heads=False)
    assert [op[0] for op in fused.plan].count("head") == 0 and [op[0] for op in plain.plan].count("head") == 2
    for a, b, r in zip(fused.forward(x), plain.forward(x), ref):
        assert a.shape[-1] in (40, 46)
        rng = float(np.abs(r).max())
        assert float((a - b).abs().max()) <= 2e-2 * rng
        assert float(np.abs(a.cpu().numpy() - r).max()) <= 3e-2 * rng


@pytest.mark.parametrize("layout", ["nhwc", "planes16"])
@pytest.mark.parametrize("B,H,W,full,pooled", [(2, 32, 64, True, False), (1, 48, 96, False, True), (2, 16, 32, True, True),
                                                (1, 37, 45, True, False), (1, 18, 34, False, True)])
def test_conv_pair_is_bitwise_two_convs(B, H, W, full, pooled, layout):
    """sa_conv3x3_pair_bf16 (16 -> 32 -> 32 encoder block, intermediate in LDS) == sa_conv3x3_bf16 twice, bit for bit:
    same bf16 rounding of the intermediate, same MFMA accumulation order; ragged tiles and image borders included."""
    from sleap_amd import _lib, ops
    from sleap_amd._lib import check
    from sleap_amd.ops import _ptr, _stream

    g = torch.Generator(device="cpu").manual_seed(H * 10 + W)
    ka = torch.randn((3, 3, 16, 32), generator=g) * (2.0 / (9 * 16)) ** 0.5
    kb = torch.randn((3, 3, 32, 32), generator=g) * (2.0 / (9 * 32)) ** 0.5
    ba, bb = (0.1 * torch.randn((32,), generator=g)).cuda(), (0.1 * torch.randn((32,), generator=g)).cuda()
    x = ops.to_bf16_padded(torch.randn((B, H, W, 16), generator=g).cuda())
    wa, wb = ops.pack_conv3x3_weights(ka.numpy(), 16), ops.pack_conv3x3_weights(kb.numpy(), 32)
    mid = ops.conv3x3(x, None, 0, wa, ba, 32, True, (H, W))
    ref = ops.conv3x3(mid, None, 0, wb, bb, 32, True, (H, W), full=full, pooled=pooled)
    ref = ref if isinstance(ref, tuple) else ((ref, None) if full else (None, ref))
    out = torch.full((B, H, W, 32), 7.0, dtype=TD, device="cuda") if full else None
    outp = torch.full((B, H // 2, W // 2, 32), 7.0, dtype=TD, device="cuda") if pooled else None
    lay = _lib.LAYOUT_PLANES16 if layout == "planes16" else _lib.LAYOUT_NHWC  # a 16-channel input is the same bytes in both
    check(_lib.lib().sa_conv3x3_pair_bf16(_ptr(x), 16, _ptr(wa), _ptr(ba), 1, 32, _ptr(wb), _ptr(bb), 1, 32, B, H, W, _ptr(out),
                                          _ptr(outp), lay, _stream()), "sa_conv3x3_pair_bf16")
    back = ops.from_planes16 if lay else (lambda t: t)
    if full:
        assert torch.equal(back(out), ref[0])
    if pooled:
        assert torch.equal(back(outp), ref[1])


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("dtype", STORAGE_DTYPES)
@pytest.mark.parametrize("layout", ["nhwc", "planes16"])
@pytest.mark.parametrize("B,H,W,full,pooled,limit", [
    (2, 32, 64, True, True, 0), (1, 48, 96, False, True, 0), (2, 16, 32, True, False, 0), (1, 37, 45, True, False, 0),
    (1, 18, 34, True, True, 0),       # ragged right / bottom tiles whose halo is mostly outside the image
    (3, 64, 160, True, True, 9),      # 60 tiles on 9 workgroups: 6-7 tiles each, XCD ranges of uneven length
    (5, 128, 128, True, True, 0),     # 160 tiles, one per workgroup
    (2, 256, 256, False, True, 0)])   # 256 tiles at the benchmark layer's size
def test_conv_pair64_is_bitwise_two_convs(B, H, W, full, pooled, limit, layout, dtype, waves, monkeypatch):
    """Round 6: the 32 -> 64 -> 64 form of sa_conv3x3_pair_bf16 (encoder block 2, csrc/convpair64.hip: one persistent workgroup
    per CU, the 64-channel intermediate only in LDS) == sa_conv3x3_bf16 twice, bit for bit -- same rounding of the intermediate,
    same MFMA accumulation order (chunk-major, taps inside); image borders, ragged tiles, several tiles per workgroup
    (sa_conv3x3_set_grid_limit), both layouts, both storage types, both workgroup shapes (SA_PAIR64_WAVES: two waves per SIMD x two
    rows -- the default -- and one wave per SIMD x four rows). Outputs are poisoned first: a skipped tile must not pass."""
    monkeypatch.setenv("SA_PAIR64_WAVES", str(waves))
    from sleap_amd import _lib, ops
    from sleap_amd._lib import check
    from sleap_amd.ops import _ptr, _stream

    td = torch.float16 if dtype == "fp16" else torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(H * 10 + W)
    ka = torch.randn((3, 3, 32, 64), generator=g) * (2.0 / (9 * 32)) ** 0.5
    kb = torch.randn((3, 3, 64, 64), generator=g) * (2.0 / (9 * 64)) ** 0.5
    ba, bb = (0.1 * torch.randn((64,), generator=g)).cuda(), (0.1 * torch.randn((64,), generator=g)).cuda()
    x = ops.to_bf16_padded(torch.randn((B, H, W, 32), generator=g).cuda(), dtype=dtype)
    wa, wb = ops.pack_conv3x3_weights(ka.numpy(), 32, dtype=dtype), ops.pack_conv3x3_weights(kb.numpy(), 64, dtype=dtype)
    mid = ops.conv3x3(x, None, 0, wa, ba, 64, True, (H, W))
    ref = ops.conv3x3(mid, None, 0, wb, bb, 64, True, (H, W), full=full, pooled=pooled)
    ref = ref if isinstance(ref, tuple) else ((ref, None) if full else (None, ref))
    out = torch.full((B, H, W, 64), float("nan"), dtype=td, device="cuda") if full else None
    outp = torch.full((B, H // 2, W // 2, 64), float("nan"), dtype=td, device="cuda") if pooled else None
    planes = layout == "planes16"
    lay = _lib.LAYOUT_PLANES16 if planes else _lib.LAYOUT_NHWC
    h = _lib.lib(dtype)
    h.sa_conv3x3_set_grid_limit(limit)
    try:
        check(h.sa_conv3x3_pair_bf16(_ptr(ops.to_planes16(x) if planes else x), 32, _ptr(wa), _ptr(ba), 1, 64, _ptr(wb), _ptr(bb), 1, 64,
                                     B, H, W, _ptr(out), _ptr(outp), lay, _stream()), "sa_conv3x3_pair_bf16")
    finally:
        h.sa_conv3x3_set_grid_limit(0)
    back = ops.from_planes16 if planes else (lambda t: t)
    if full:
        assert torch.equal(back(out), ref[0])
    if pooled:
        assert torch.equal(back(outp), ref[1])


@pytest.mark.parametrize("C0,C1,Cout,B,H,W,full,pooled", [
    (32, 0, 64, 2, 32, 64, True, True), (64, 0, 64, 1, 48, 96, False, True), (64, 128, 64, 2, 16, 32, True, False),
    (48, 0, 128, 1, 37, 45, True, False), (128, 256, 128, 1, 18, 34, True, False), (16, 0, 32, 2, 20, 40, True, True),
    (32, 0, 32, 1, 24, 40, True, False)])
def test_conv3x3_planes16_is_bitwise_nhwc(C0, C1, Cout, B, H, W, full, pooled):
    """SA_LAYOUT_PLANES16 changes where the bytes lie, not what they are: sa_conv3x3_bf16 on 16-channel planes (sources,
    full-resolution and pooled outputs) == the NHWC call, bit for bit -- single- and multi-chunk kernels, concatenated
    sources, ragged tiles. The one exception is the 32 -> (<= 32) layer, which NHWC runs as ONE 32-channel chunk and planes
    as two 16-channel chunks (another fp32 summation order): equal to within one rounding of the storage type."""
    from sleap_amd import _lib, ops

    g = torch.Generator(device="cpu").manual_seed(C0 + 3 * C1 + H)
    k = (torch.randn((3, 3, C0 + C1, Cout), generator=g) * (2.0 / (9 * (C0 + C1))) ** 0.5).numpy()
    pw = ops.pack_conv3x3_weights(k, C0, C1)
    bias = (0.1 * torch.randn((Cout,), generator=g)).cuda()
    x0 = ops.to_bf16_padded(torch.randn((B, H, W, C0), generator=g).cuda())
    x1 = ops.to_bf16_padded(torch.randn((B, H, W, C1), generator=g).cuda()) if C1 else None
    mode = 1 if C1 else 0
    a = ops.conv3x3(x0, x1, mode, pw, bias, Cout, True, (H, W), full=full, pooled=pooled)
    b = ops.conv3x3(ops.to_planes16(x0), ops.to_planes16(x1) if C1 else None, mode | _lib.LAYOUT_PLANES16, pw, bias, Cout, True,
                    (H, W), full=full, pooled=pooled)
    a = a if isinstance(a, tuple) else (a,)
    b = b if isinstance(b, tuple) else (b,)
    for u, v in zip(a, b):
        if C0 + C1 == 32 and Cout <= 32:
            assert float((u.float() - ops.from_planes16(v).float()).abs().max()) <= 2.0 ** -8 * float(u.float().abs().max())
        else:
            assert torch.equal(u, ops.from_planes16(v))


@pytest.mark.parametrize("C0,C1,Cout,B,H,W,full,pooled", [
    (64, 128, 64, 2, 32, 64, True, False), (256, 512, 256, 1, 16, 32, True, False), (128, 256, 128, 1, 48, 96, True, True),
    (16, 32, 32, 2, 18, 34, True, False), (64, 64, 128, 1, 12, 20, True, False), (32, 48, 16, 1, 64, 32, True, False),
    (64, 128, 64, 1, 2, 2, True, False)])
def test_conv3x3_upsampling_source_on_planes_is_the_materialised_path(C0, C1, Cout, B, H, W, full, pooled):
    """SA_SRC1_UPSAMPLE2X on 16-channel planes (the half-resolution tile expanded in LDS) vs sa_upsample2x_bf16 followed by the
    SA_SRC1_DIRECT convolution. bf16 build: bit for bit (same fp32 interpolation arithmetic, one rounding, same accumulation
    order). fp16 build (round 3): the expansion interpolates in packed fp16 (horizontal blend rounded before the vertical
    one), i.e. the expanded operand differs from the materialised tensor by <= 1.5 fp16 ulp -- outputs within 2e-3 of the
    output range, and both within the same distance of an fp32 interpolation (test_fused_upsampling_network_...). Ragged
    tiles, image borders (clamped sources, zero padding of the halo), one and two 32-channel output tiles per workgroup."""
    from sleap_amd import _lib, ops
    from sleap_amd._lib import check
    from sleap_amd.ops import _ptr, _stream

    g = torch.Generator(device="cpu").manual_seed(C0 + 3 * C1 + H)
    k = (torch.randn((3, 3, C0 + C1, Cout), generator=g) * (2.0 / (9 * (C0 + C1))) ** 0.5).numpy()
    pw = ops.pack_conv3x3_weights(k, C0, C1)
    bias = (0.1 * torch.randn((Cout,), generator=g)).cuda()
    x0 = ops.to_planes16(ops.to_bf16_padded(torch.randn((B, H, W, C0), generator=g).cuda()))
    low = ops.to_planes16(ops.to_bf16_padded(torch.randn((B, H // 2, W // 2, C1), generator=g).cuda()))
    up = torch.empty((B, H, W, low.shape[3]), dtype=TD, device="cuda")
    check(_lib.lib().sa_upsample2x_bf16(_ptr(low), B * (low.shape[3] // 16), H // 2, W // 2, 16, 1, _ptr(up), _stream()), "sa_upsample2x_bf16")
    P = _lib.LAYOUT_PLANES16
    a = ops.conv3x3(x0, up, _lib.SRC1_DIRECT | P, pw, bias, Cout, True, (H, W), full=full, pooled=pooled)
    b = ops.conv3x3(x0, low, _lib.SRC1_UPSAMPLE2X | P, pw, bias, Cout, True, (H, W), full=full, pooled=pooled)
    a = a if isinstance(a, tuple) else (a,)
    b = b if isinstance(b, tuple) else (b,)
    for u, v in zip(a, b):
        if _lib.DEFAULT_DTYPE == "fp16":
            assert float((u.float() - v.float()).abs().max()) <= 2e-3 * float(u.float().abs().max())
        else:
            assert torch.equal(u, v)


def test_fused_upsampling_network_is_the_materialised_one():
    """DeviceNetwork(fuse_upsample=True) on planes: no upsampling launches. bf16 build: same outputs bit for bit; fp16 build
    (packed-fp16 expansion): heads within 2e-3 of their range of the materialised network AND no further from the fp32 oracle
    than the materialised network is. C executor and the per-launch Python loop issue the same launches (bitwise)."""
    from oracle.keras_graph import KerasGraph, ensure_float
    from sleap_amd import _lib
    from sleap_amd.nn.engine import DeviceNetwork

    cfg, w = _benchmark_unet(96, 128)
    frames = _fly_frames(3, 96, 128, 11)
    x = torch.from_numpy(frames).cuda()
    a, b = DeviceNetwork(cfg, w, fuse_upsample=False), DeviceNetwork(cfg, w, fuse_upsample=True)
    assert a.planar and b.planar and sum(op[0] == "up" for op in a.plan) == 3 and not any(op[0] == "up" for op in b.plan)
    if _lib.DEFAULT_DTYPE == "fp16":  # the default: per layer -- only the conv with <= 64 output channels reads at half resolution
        assert sum(op[0] == "up" for op in DeviceNetwork(cfg, w).plan) == 2
    base = [o.clone() for o in a.forward(x)]
    fused = [o.clone() for o in b.forward(x)]
    if _lib.DEFAULT_DTYPE == "fp16":
        ref = KerasGraph(cfg, w)(ensure_float(frames))
        for p, q, r in zip(base, fused, ref):
            rng = float(np.abs(r).max())
            assert float((p - q).abs().max()) <= 2e-3 * rng
            ea, eb = float(np.abs(p.cpu().numpy() - r).max()), float(np.abs(q.cpu().numpy() - r).max())
            print(f"vs fp32 oracle: materialised {ea / rng:.2e}, fused {eb / rng:.2e} of the range")
            assert eb <= max(1.5 * ea, 2e-3 * rng)
    else:
        for p, q in zip(base, fused):
            assert torch.equal(p, q)
    prof = []
    for p, q in zip(fused, b.forward(x, profile=prof)):
        assert torch.equal(p, q)


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 12, 16), (1, 5, 7, 48), (3, 16, 16, 128)])
def test_upsample2x_planes16_is_bitwise_nhwc_and_matches_torch(B, H, W, C):
    """UpSampling2D(2, bilinear) (encoder_decoder.py:335-339): the 16-channel kernel that serves plane tensors writes the same
    bits as the NHWC kernel, and both are torch's half-pixel bilinear interpolation up to the storage rounding."""
    from sleap_amd import _lib, ops
    from sleap_amd._lib import check
    from sleap_amd.ops import _ptr, _stream

    g = torch.Generator(device="cpu").manual_seed(B + H + C)
    xf = torch.randn((B, H, W, C), generator=g).cuda()
    x = ops.to_bf16_padded(xf)
    out = torch.empty((B, 2 * H, 2 * W, C), dtype=TD, device="cuda")
    check(_lib.lib().sa_upsample2x_bf16(_ptr(x), B, H, W, C, 1, _ptr(out), _stream()), "sa_upsample2x_bf16")
    outp = torch.empty_like(out)
    check(_lib.lib().sa_upsample2x_bf16(_ptr(ops.to_planes16(x)), B * (C // 16), H, W, 16, 1, _ptr(outp), _stream()), "sa_upsample2x_bf16")
    assert torch.equal(out, ops.from_planes16(outp))
    ref = torch.nn.functional.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False)
    assert float((out.float() - ref.permute(0, 2, 3, 1)).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())


def test_network_layout_is_bitwise_neutral():
    """The benchmark UNet compiled on 16-channel planes (the default for the UNet family) and forced to NHWC: identical heads;
    a plan with an un-fused head (fuse_heads=False) or an upsample-on-load conv stays NHWC on its own."""
    from sleap_amd.nn.engine import DeviceNetwork

    cfg, w = _benchmark_unet(96, 128)
    x = torch.from_numpy(_fly_frames(2, 96, 128, 7)).cuda()
    a = DeviceNetwork(cfg, w, fuse_upsample=False)  # (the per-layer default expands one upsampling in LDS, on planes only)
    b = DeviceNetwork(cfg, w, layout="nhwc")
    assert a.planar and not b.planar
    assert DeviceNetwork(cfg, w, fuse_upsample=True).planar
    # round 3: un-fused 1x1 heads read planes too (the matrix-core head kernel) -- same bits as the fused heads' NHWC twin
    c = DeviceNetwork(cfg, w, fuse_heads=False, fuse_upsample=False)
    d = DeviceNetwork(cfg, w, fuse_heads=False, fuse_upsample=False, layout="nhwc")
    assert c.planar and not d.planar
    for p_, q_ in zip([o.clone() for o in c.forward(x)], d.forward(x)):
        assert torch.equal(p_, q_)
    base = [o.clone() for o in a.forward(x)]
    for p, q in zip(base, b.forward(x)):
        assert torch.equal(p, q)
    prof = []
    for p, q in zip(base, a.forward(x, profile=prof)):  # the per-launch Python loop (bench.py --layers) passes the layout too
        assert torch.equal(p, q)


def test_pair_fusion_switch_is_bitwise_neutral():
    from sleap_amd.nn.engine import DeviceNetwork

    cfg, w = _benchmark_unet(96, 128)
    x = torch.from_numpy(_fly_frames(2, 96, 128, 6)).cuda()
    # NHWC on both sides: the fused kernel reproduces the un-fused NHWC arithmetic (the 32 -> 32 layer as ONE 32-channel chunk);
    # on planes the un-fused layer runs as two 16-channel chunks (test_conv3x3_planes16_is_bitwise_nhwc)
    a = DeviceNetwork(cfg, w, layout="nhwc")
    assert "pair" in [op[0] for op in a.plan]
    base = [o.clone() for o in a.forward(x)]
    other = [o.clone() for o in DeviceNetwork(cfg, w, fuse_pairs=False, layout="nhwc").forward(x)]
    for p, q in zip(base, other):
        assert torch.equal(p, q)


def test_block2_fusion_is_bitwise_neutral_in_the_network_on_both_sides_of_its_size_rule(monkeypatch):
    """Round 6: the plan of the benchmark UNet holds TWO "pair" ops (16 -> 32 -> 32 and 32 -> 64 -> 64). The second one is launched
    fused from ~6 tiles per CU on and as two convolutions through its kept intermediate buffer below that (engine.py:_fuse_pairs,
    csrc/network.hip K_PAIR) -- heads bitwise equal to the plan without it (SA_FUSE_PAIRS64=0) on both sides of the rule, through
    the C executor and through the per-launch Python loop."""
    from sleap_amd.nn.engine import DeviceNetwork

    cfg, w = _benchmark_unet(512, 512)
    g = torch.Generator(device="cpu").manual_seed(3)
    for B in (2, 56):  # 2 x 32 = 64 tiles of the 128 x 128 layer (two launches); 56 x 32 = 1792 >= 6 x 256 (the fused launch)
        x = torch.randint(0, 256, (B, 512, 512, 1), generator=g, dtype=torch.uint8).cuda()
        monkeypatch.delenv("SA_FUSE_PAIRS64", raising=False)
        a = DeviceNetwork(cfg, w)
        assert [op[0] for op in a.plan].count("pair") == 2
        base = [o.clone() for o in a.forward(x)]
        prof = []
        for p, q in zip(base, a.forward(x, profile=prof)):
            assert torch.equal(p, q)
        monkeypatch.setenv("SA_FUSE_PAIRS64", "0")
        b = DeviceNetwork(cfg, w)
        assert [op[0] for op in b.plan].count("pair") == 1
        for p, q in zip(base, b.forward(x)):
            assert torch.equal(p, q)
        del a, b
