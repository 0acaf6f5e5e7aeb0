"""Hourglass backbone on the device (SURVEY.md §8a row a2'): the extended conv epilogue (BatchNormalization after
ReLU, residual Add with nearest-neighbour upsampling), the general first-layer conv, and whole-graph parity of
`build_hourglass_model_config` graphs against the CPU oracle (oracle/keras_graph.py).

Layer tests compare with a plain torch fp32 evaluation of the same op on the same bf16-rounded operands
(|delta| <= 1e-2 max|ref|: accumulation order + one final bf16 rounding)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# the suite follows the storage type of the library build under test (SLEAP_AMD_DTYPE, default fp16); the tolerances below were
# written for bf16 (8 mantissa bits) and hold a fortiori for fp16 (11)
from sleap_amd import _lib as _L  # noqa: E402

TD = {"bf16": torch.bfloat16, "fp16": torch.float16}[_L.DEFAULT_DTYPE]


def _bf(x):
    return x.to(TD).to(torch.float32)


def _padded(v, n, fill=0.0):
    out = torch.full((n,), fill, dtype=torch.float32)
    out[: v.numel()] = v
    return out.cuda()


EX_CASES = [
    # (B, H, W, C0, C1, Cout, affine, residual: 0 none / 1 same res / 2 half res nearest, relu, relu_last, pooled)
    (2, 16, 32, 32, 0, 32, True, 0, True, False, False),
    (1, 24, 40, 64, 0, 96, True, 2, True, False, False),
    (1, 20, 36, 48, 0, 80, True, 1, True, False, False),
    (2, 16, 32, 16, 0, 16, True, 0, True, False, True),   # BN then fused max pool
    (1, 18, 34, 32, 32, 64, False, 1, False, True, False),  # ResNet style: linear conv + shortcut, ReLU last
    (1, 16, 16, 256, 0, 320, True, 2, True, False, False),
]


@pytest.mark.parametrize("B,H,W,C0,C1,Cout,affine,residual,relu,relu_last,pooled", EX_CASES)
def test_conv3x3_extended_epilogue(B, H, W, C0, C1, Cout, affine, residual, relu, relu_last, pooled):
    from sleap_amd import _lib, ops
    from sleap_amd._lib import check
    from sleap_amd.ops import _ptr, _stream

    g = torch.Generator(device="cpu").manual_seed(H * 100 + C0 + Cout + residual)
    k = torch.randn((3, 3, C0 + C1, Cout), generator=g) * (2.0 / (9 * (C0 + C1))) ** 0.5
    bias = torch.randn((Cout,), generator=g) * 0.1
    x0 = torch.randn((B, H, W, C0), generator=g)
    x1 = torch.randn((B, H, W, C1), generator=g) if C1 else None
    scale = 1.0 + 0.3 * torch.randn((Cout,), generator=g)  # negative values matter for BN-before-pool
    scale[::7] *= -1.0
    shift = 0.2 * torch.randn((Cout,), generator=g)
    res = None
    if residual == 1:
        res = torch.randn((B, H, W, Cout), generator=g)
    elif residual == 2:
        res = torch.randn((B, H // 2, W // 2, Cout), generator=g)
    # reference
    rin = _bf(x0) if x1 is None else torch.cat([_bf(x0), _bf(x1)], dim=-1)
    y = F.conv2d(rin.permute(0, 3, 1, 2), _bf(k).permute(3, 2, 0, 1), bias, padding=1).permute(0, 2, 3, 1)
    if relu:
        y = torch.relu(y)
    if affine:
        y = y * scale + shift
    if res is not None:
        r = _bf(res)
        if residual == 2:
            r = r.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
        y = y + r
    if relu_last:
        y = torch.relu(y)
    # device
    coutp = ops.pad16(Cout)
    d0 = ops.to_bf16_padded(x0.cuda().contiguous())
    d1 = ops.to_bf16_padded(x1.cuda().contiguous()) if x1 is not None else None
    dres = ops.to_bf16_padded(res.cuda().contiguous()) if res is not None else None
    pw = ops.pack_conv3x3_weights(k.numpy(), C0, C1)
    bp = _padded(bias, coutp)
    ps = _padded(scale, coutp, 1.0) if affine else None
    pt = _padded(shift, coutp) if affine else None
    out = torch.empty((B, H, W, coutp), dtype=TD, device="cuda")
    outp = torch.empty((B, H // 2, W // 2, coutp), dtype=TD, device="cuda") if pooled else None
    check(_lib.lib().sa_conv3x3_ex_bf16(_ptr(d0), d0.shape[3], _ptr(d1), d1.shape[3] if d1 is not None else 0,
                                        _lib.SRC1_DIRECT if d1 is not None else _lib.SRC1_NONE, _ptr(pw), _ptr(bp), coutp,
                                        int(relu), B, H, W, _ptr(out), _ptr(outp), _ptr(ps), _ptr(pt), _ptr(dres),
                                        1 if residual == 2 else 0, int(relu_last), _stream()), "sa_conv3x3_ex_bf16")
    got = ops.from_bf16(out, Cout).cpu()
    lim = 1e-2 * float(y.abs().max())
    assert float((got - y).abs().max()) <= lim, (float((got - y).abs().max()), lim)
    if pooled:
        want = F.max_pool2d(out.float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
        assert torch.equal(outp.float(), want)


@pytest.mark.parametrize("dtype,B,H,W,Cin,Cout,k,stride,relu,affine", [
    ("u8", 2, 64, 96, 1, 16, 7, 2, True, True),     # hourglass stem
    ("u8", 1, 50, 38, 3, 24, 7, 2, True, True),     # even sizes, RGB, channel padding
    ("f32", 1, 33, 47, 1, 16, 3, 1, True, False),
    ("f32", 2, 32, 32, 1, 32, 5, 1, False, True),
    ("u8", 1, 40, 40, 1, 8, 1, 1, True, False),
])
def test_image_conv_vs_torch(dtype, B, H, W, Cin, Cout, k, stride, relu, affine):
    """General first-layer conv with TF 'SAME' padding (pad_before = pad_total // 2), fp32 arithmetic."""
    from sleap_amd import _lib, ops
    from sleap_amd._lib import check
    from sleap_amd.ops import _ptr, _stream

    g = torch.Generator(device="cpu").manual_seed(H + W + k)
    if dtype == "u8":
        img = torch.randint(0, 256, (B, H, W, Cin), generator=g, dtype=torch.uint8)
        xf = img.float() * (1.0 / 255.0)
    else:
        img = torch.rand((B, H, W, Cin), generator=g)
        xf = img
    kern = torch.randn((k, k, Cin, Cout), generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    bias = 0.1 * torch.randn((Cout,), generator=g)
    scale = 1.0 + 0.3 * torch.randn((Cout,), generator=g)
    shift = 0.2 * torch.randn((Cout,), generator=g)
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph, pw_ = max((Ho - 1) * stride + k - H, 0), max((Wo - 1) * stride + k - W, 0)
    xp = F.pad(xf.permute(0, 3, 1, 2), (pw_ // 2, pw_ - pw_ // 2, ph // 2, ph - ph // 2))
    y = F.conv2d(xp, kern.permute(3, 2, 0, 1), bias, stride=stride).permute(0, 2, 3, 1)
    if relu:
        y = torch.relu(y)
    if affine:
        y = y * scale + shift
    assert y.shape[1:3] == (Ho, Wo)
    coutp = ops.pad16(Cout)
    w = torch.zeros((k, k, Cin, coutp))
    w[..., :Cout] = kern
    dimg, dw, db = img.cuda().contiguous(), w.cuda().contiguous(), _padded(bias, coutp)
    ps = _padded(scale, coutp, 1.0) if affine else None
    pt = _padded(shift, coutp) if affine else None
    out = torch.empty((B, Ho, Wo, coutp), dtype=TD, device="cuda")
    check(_lib.lib().sa_image_conv_bf16(_ptr(dimg), 1 if dtype == "u8" else 0, B, H, W, Cin, Cin, None, k, k, stride, ph // 2,
                                        pw_ // 2,
                                        Ho, Wo, _ptr(dw), _ptr(db), coutp, int(relu), _ptr(ps), _ptr(pt), _ptr(out),
                                        _stream()), "sa_image_conv_bf16")
    got = ops.from_bf16(out, Cout).cpu()
    # fp32 arithmetic on both sides: only the final bf16 rounding (2^-9 relative) separates them
    assert float((got - y).abs().max()) <= 2.0 ** -8 * float(y.abs().max())
    assert float(out.float()[..., Cout:].abs().max()) == 0.0 if coutp > Cout else True


@pytest.mark.parametrize("half,relu", [(0, 0), (1, 0), (1, 1)])
def test_add_vs_torch(half, relu):
    from sleap_amd import _lib, ops
    from sleap_amd._lib import check
    from sleap_amd.ops import _ptr, _stream

    g = torch.Generator(device="cpu").manual_seed(7 + half)
    a = torch.randn((2, 12, 20, 48), generator=g)
    b = torch.randn((2, 6, 10, 48) if half else (2, 12, 20, 48), generator=g)
    da, db = ops.to_bf16_padded(a.cuda()), ops.to_bf16_padded(b.cuda())
    out = torch.empty_like(da)
    check(_lib.lib().sa_add_bf16(_ptr(da), _ptr(db), 2, 12, 20, 48, half, relu, _ptr(out), _stream()), "sa_add_bf16")
    rb = _bf(b)
    if half:
        rb = rb.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    ref = _bf(a) + rb
    if relu:
        ref = torch.relu(ref)
    assert torch.equal(out.float().cpu(), _bf(ref))


def _hourglass(h, w, stacks, heads, interp="nearest", seed=1, cin=1):
    from sleap_amd.nn.architectures import build_hourglass_model_config, he_normal_weights

    cfg, shapes = build_hourglass_model_config((h, w, cin), stem_stride=4, max_stride=32, output_stride=4, stem_filters=16,
                                               filters=32, filter_increase=16, stacks=stacks, interp_method=interp,
                                               heads=heads)
    return cfg, he_normal_weights(shapes, seed=seed)


def _parity(cfg, w, x_u8, tol_fp32, tol_bf16):
    from oracle.keras_graph import KerasGraph, ensure_float
    from sleap_amd.nn.engine import DeviceNetwork

    net = DeviceNetwork(cfg, w)
    outs = [o.cpu().numpy() for o in net.forward(torch.from_numpy(x_u8).cuda())]
    xin = ensure_float(x_u8)
    res = {}
    for mode, tol in (("fp32", tol_fp32), ("bf16", tol_bf16)):
        ref = KerasGraph(cfg, w, emulate_bf16=(mode == "bf16"), round_layers=net.round_points)(xin)
        assert len(ref) == len(outs)
        worst = 0.0
        for o, r in zip(outs, ref):
            assert o.shape == r.shape
            assert np.isfinite(o).all() and np.isfinite(r).all()  # max() below would silently drop a NaN
            worst = max(worst, float(np.abs(o - r).max() / np.abs(r).max()))
        assert worst <= tol, f"{mode}: {worst:.4g} > {tol}"
        res[mode] = worst
    return net, res


def test_hourglass_heads_vs_oracle():
    """1-stack hourglass (stem k7 s2 + BN everywhere + nearest-upsample Add skips) with the bottom-up heads:
    max|delta|/max|ref| <= 3e-2 vs the fp32 oracle and <= 2e-2 vs the oracle rounding to bf16 at the engine's
    storage points. The whole decoder runs without a standalone upsample / add / pool launch."""
    cfg, w = _hourglass(128, 160, 1, [("MultiInstanceConfmapsHead", 13, 4), ("PartAffinityFieldsHead", 24, 4)])
    rng = np.random.default_rng(0)
    net, _ = _parity(cfg, w, rng.integers(0, 256, (2, 128, 160, 1), dtype=np.uint8), 3e-2, 2e-2)
    kinds = [op[0] for op in net.plan]
    assert "add" not in kinds and "up" not in kinds and "pool" not in kinds, kinds


@pytest.mark.parametrize("interp,cin", [("nearest", 1), ("bilinear", 3)])
def test_hourglass_plan_on_planes_is_the_nhwc_plan_bit_for_bit(interp, cin):
    """Round 3: hourglass plans compile to 16-channel planes (k7 first-layer convolution storing planes, 3x3 convolutions with
    the BatchNormalization / residual epilogue, un-fused heads reading planes). Same launches, same arithmetic, another
    address map: the NHWC twin of the plan gives identical bits."""
    from sleap_amd.nn.engine import DeviceNetwork

    from sleap_amd.nn.architectures import build_hourglass_model_config, he_normal_weights

    # (no 32 -> <= 32 channel convolution in this one: that shape takes a single 32-channel chunk on NHWC tensors and two
    #  16-channel chunks on planes -- another accumulation order, the one legitimate difference between the two plans)
    cfg, shapes = build_hourglass_model_config((128, 160, cin), stem_stride=4, max_stride=32, output_stride=4, stem_filters=32, filters=64,
                                               filter_increase=32, stacks=1, interp_method=interp,
                                               heads=[("MultiInstanceConfmapsHead", 13, 4), ("PartAffinityFieldsHead", 24, 4)])
    w = he_normal_weights(shapes, seed=1)
    x = torch.from_numpy(np.random.default_rng(5).integers(0, 256, (3, 128, 160, cin), dtype=np.uint8)).cuda()
    a, b = DeviceNetwork(cfg, w), DeviceNetwork(cfg, w, layout="nhwc")
    assert a.planar and not b.planar
    for p_, q_ in zip([o.clone() for o in a.forward(x)], b.forward(x)):
        assert torch.equal(p_, q_)
    # float32 frames take the other first-layer kernel (sa_image_conv_bf16 instead of the matrix-core u8 one): its plane store
    xf = x.float() / 255.0  # ensure_float: float frames are taken as they are, uint8 ones are scaled by 1/255
    ref = [o.clone() for o in a.forward(x)]
    for p_, q_, r_ in zip([o.clone() for o in a.forward(xf)], b.forward(xf), ref):
        assert torch.equal(p_, q_)
        assert float((p_ - r_).abs().max()) <= 2e-2 * float(r_.abs().max())


def test_hourglass_stacked_features_vs_oracle():
    """2-stack hourglass without heads (what the reference's own architecture test builds): both stack outputs."""
    cfg, w = _hourglass(96, 96, 2, [])
    rng = np.random.default_rng(1)
    net, _ = _parity(cfg, w, rng.integers(0, 256, (1, 96, 96, 1), dtype=np.uint8), 4e-2, 2e-2)
    assert len(net.outputs) == 2


def test_hourglass_bilinear_rgb_vs_oracle():
    """UpsamplingBlock's other interpolation (hourglass.py:160) + RGB input: bilinear upsample materialised,
    the Add still folds into the skip conv (same-resolution residual)."""
    cfg, w = _hourglass(64, 96, 1, [("SingleInstanceConfmapsHead", 5, 4)], interp="bilinear", cin=3)
    rng = np.random.default_rng(2)
    net, _ = _parity(cfg, w, rng.integers(0, 256, (2, 64, 96, 3), dtype=np.uint8), 3e-2, 2e-2)
    kinds = [op[0] for op in net.plan]
    assert "add" not in kinds and "up" in kinds


# ------------------------------------------------------------------------------------------------
# tap-GEMM kernels (ResNet 1x1 convs, transposed convs) and the general max pool
# ------------------------------------------------------------------------------------------------
def _pack_taps(w_taps, cinp, coutp):
    """w_taps (T, Cin, Cout) f32 torch -> packed device tensor."""
    from sleap_amd import _lib
    from sleap_amd._lib import check
    import ctypes as C

    T, cin, cout = w_taps.shape
    h = _lib.lib()
    n = h.sa_tapconv_packed_elems(T, cinp, coutp)
    packed = np.zeros((n,), np.uint16)
    wn = np.ascontiguousarray(w_taps.numpy(), dtype=np.float32)
    check(h.sa_pack_tapconv_weights(wn.ctypes.data_as(C.c_void_p), T, cin, cinp, cout, coutp, packed.ctypes.data_as(C.c_void_p)),
          "sa_pack_tapconv_weights")
    return torch.from_numpy(packed.view(np.int16)).cuda()


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,affine,residual,relu,relu_last", [
    (2, 16, 24, 64, 64, 1, True, False, False, True),     # block _1_conv + bn + relu
    (1, 20, 28, 64, 256, 1, True, True, False, True),     # block _3_conv + bn + add + relu
    (2, 32, 32, 256, 128, 2, True, False, False, True),   # strided block1 _1_conv
    (1, 17, 23, 48, 80, 2, False, False, True, False),    # ragged tiles, channel padding, odd size
    (1, 8, 8, 512, 2048, 1, True, False, False, False),   # shortcut conv
    (1, 40, 40, 2048, 32, 1, False, False, False, False),
])
def test_conv1x1_vs_torch(B, H, W, Cin, Cout, stride, affine, residual, relu, relu_last):
    from sleap_amd import _lib, ops
    from sleap_amd._lib import check
    from sleap_amd.ops import _ptr, _stream

    g = torch.Generator(device="cpu").manual_seed(H + Cin + Cout)
    k = torch.randn((Cin, Cout), generator=g) * (2.0 / Cin) ** 0.5
    bias = 0.1 * torch.randn((Cout,), generator=g)
    x = torch.randn((B, H, W, Cin), generator=g)
    scale = 1.0 + 0.3 * torch.randn((Cout,), generator=g)
    shift = 0.2 * torch.randn((Cout,), generator=g)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = torch.randn((B, Ho, Wo, Cout), generator=g) if residual else None
    y = torch.einsum("bhwc,cd->bhwd", _bf(x)[:, ::stride, ::stride], _bf(k)) + bias
    if relu:
        y = torch.relu(y)
    if affine:
        y = y * scale + shift
    if res is not None:
        y = y + _bf(res)
    if relu_last:
        y = torch.relu(y)
    cinp, coutp = ops.pad16(Cin), ops.pad16(Cout)
    dx = ops.to_bf16_padded(x.cuda().contiguous())
    dres = ops.to_bf16_padded(res.cuda().contiguous()) if res is not None else None
    pw = _pack_taps(k[None], cinp, coutp)
    bp = _padded(bias, coutp)
    ps = _padded(scale, coutp, 1.0) if affine else None
    pt = _padded(shift, coutp) if affine else None
    out = torch.full((B, Ho, Wo, coutp), 7.0, dtype=TD, device="cuda")
    check(_lib.lib().sa_conv1x1_bf16(_ptr(dx), cinp, _ptr(pw), _ptr(bp), coutp, int(relu), B, H, W, stride, _ptr(ps), _ptr(pt),
                                     _ptr(dres), int(relu_last), _ptr(out), _stream()), "sa_conv1x1_bf16")
    got = ops.from_bf16(out, Cout).cpu()
    lim = 1e-2 * float(y.abs().max())
    assert float((got - y).abs().max()) <= lim, (float((got - y).abs().max()), lim)
    if coutp > Cout:
        pad = out.float()[..., Cout:]
        assert float(pad.abs().max()) == 0.0 


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,residual", [
    (2, 24, 20, 64, 256, 1, False),    # the cout-tile-walking kernel (stride 1, 64-128 input channels, >= 256 output channels)
    (1, 19, 23, 128, 512, 1, True),    # + residual, ragged pixel tile
    (3, 16, 16, 96, 272, 1, True),     # three K chunks of 32, a ragged last cout tile (272 = 2 x 128 + 16)
    (1, 12, 12, 256, 1024, 1, True),   # 256 input channels: the per-(pixel tile, cout tile) kernel
    (1, 24, 24, 128, 512, 2, False),   # stride 2: likewise
])
def test_conv1x1_on_planes_is_bitwise_nhwc(B, H, W, Cin, Cout, stride, residual):
    """The 1 x 1 convolution on 16-channel planes -- for ResNet's expand shapes a kernel of its own (tapconv.hip:
    tapconv_coloop_kernel: the pixel tile resident in LDS, the cout tiles walked by one workgroup) -- gives the bits of the NHWC
    launch: same products, same accumulation order, same epilogue."""
    from sleap_amd import _lib, ops
    from sleap_amd._lib import check
    from sleap_amd.ops import _ptr, _stream

    g = torch.Generator(device="cpu").manual_seed(B + H + Cin + Cout)
    k = torch.randn((Cin, Cout), generator=g) * (2.0 / Cin) ** 0.5
    x = torch.randn((B, H, W, Cin), generator=g)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    cinp, coutp = ops.pad16(Cin), ops.pad16(Cout)
    dx = ops.to_bf16_padded(x.cuda().contiguous())
    dres = ops.to_bf16_padded(torch.randn((B, Ho, Wo, Cout), generator=g).cuda().contiguous()) if residual else None
    pw = _pack_taps(k[None], cinp, coutp)
    bp = _padded(0.1 * torch.randn((Cout,), generator=g), coutp)
    ps = _padded(1.0 + 0.3 * torch.randn((Cout,), generator=g), coutp, 1.0)
    pt = _padded(0.2 * torch.randn((Cout,), generator=g), coutp)
    outs = []
    for lay in (_lib.LAYOUT_NHWC, _lib.LAYOUT_PLANES16):
        pl = lay == _lib.LAYOUT_PLANES16
        src = ops.to_planes16(dx) if pl else dx
        res = (ops.to_planes16(dres) if pl else dres) if residual else None
        out = torch.full((B, Ho, Wo, coutp), 7.0, dtype=TD, device="cuda")
        check(_lib.lib().sa_conv1x1_bf16(_ptr(src), cinp, _ptr(pw), _ptr(bp), coutp, lay, B, H, W, stride, _ptr(ps), _ptr(pt),
                                         _ptr(res), 1, _ptr(out), _stream()), "sa_conv1x1_bf16")
        outs.append(ops.from_planes16(out) if pl else out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    assert float(outs[0].float().abs().max()) > 0.5


@pytest.mark.parametrize("B,H,W,Cin,Cout,ksize,affine", [
    (2, 8, 12, 64, 64, 4, True),
    (1, 9, 7, 48, 80, 4, False),
    (1, 16, 16, 256, 128, 4, True),
    (2, 10, 6, 32, 24, 3, False),
])
def test_convt_s2_vs_torch(B, H, W, Cin, Cout, ksize, affine):
    """Conv2DTranspose(k, s2, same): TF crops the full transposed conv at offset max(k - 2, 0) // 2."""
    import ctypes as C

    from sleap_amd import _lib, ops
    from sleap_amd._lib import check
    from sleap_amd.ops import _ptr, _stream

    g = torch.Generator(device="cpu").manual_seed(H * 7 + Cin + ksize)
    kt = torch.randn((ksize, ksize, Cout, Cin), generator=g) * (2.0 / (Cin * 4)) ** 0.5  # Keras layout
    bias = 0.1 * torch.randn((Cout,), generator=g)
    x = torch.randn((B, H, W, Cin), generator=g)
    scale = 1.0 + 0.3 * torch.randn((Cout,), generator=g)
    shift = 0.2 * torch.randn((Cout,), generator=g)
    full = F.conv_transpose2d(_bf(x).permute(0, 3, 1, 2), _bf(kt).permute(3, 2, 0, 1), None, stride=2)
    c = max(ksize - 2, 0) // 2
    y = full[:, :, c:c + 2 * H, c:c + 2 * W].permute(0, 2, 3, 1) + bias
    if affine:
        y = y * scale + shift
    y = torch.relu(y)
    cinp, coutp = ops.pad16(Cin), ops.pad16(Cout)
    h = _lib.lib()
    phases = []
    for ph in range(4):
        ky, kx = (C.c_int * 4)(), (C.c_int * 4)()
        n = h.sa_convt_s2_phase_taps(ksize, ph, ky, kx)
        taps = torch.stack([kt[ky[i], kx[i]].T for i in range(n)])  # (n, Cin, Cout)
        phases.append(_pack_taps(taps.contiguous(), cinp, coutp))
    arr = (C.c_void_p * 4)(*[t.data_ptr() for t in phases])
    dx = ops.to_bf16_padded(x.cuda().contiguous())
    bp = _padded(bias, coutp)
    ps = _padded(scale, coutp, 1.0) if affine else None
    pt = _padded(shift, coutp) if affine else None
    out = torch.empty((B, 2 * H, 2 * W, coutp), dtype=TD, device="cuda")
    # BN sits between the transposed conv and the ReLU (upsampling.py:186-188): relu = 0, affine, relu_last = 1
    check(h.sa_convt_s2_bf16(_ptr(dx), cinp, arr, ksize, _ptr(bp), coutp, 0, B, H, W, _ptr(ps), _ptr(pt), 1, _ptr(out),
                             _stream()), "sa_convt_s2_bf16")
    got = ops.from_bf16(out, Cout).cpu()
    lim = 1e-2 * float(y.abs().max())
    assert float((got - y).abs().max()) <= lim, (float((got - y).abs().max()), lim)


@pytest.mark.parametrize("k,stride,pad,pad_zero,H,W", [(3, 2, 1, 1, 32, 48), (2, 2, 0, 0, 16, 16), (3, 2, 1, 0, 17, 21)])
def test_maxpool_general_vs_torch(k, stride, pad, pad_zero, H, W):
    from sleap_amd import _lib, ops
    from sleap_amd._lib import check
    from sleap_amd.ops import _ptr, _stream

    g = torch.Generator(device="cpu").manual_seed(k + H)
    x = torch.randn((2, H, W, 32), generator=g) - (1.0 if pad_zero else 0.0)
    dx = ops.to_bf16_padded(x.cuda())
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((2, Ho, Wo, 32), dtype=TD, device="cuda")
    check(_lib.lib().sa_maxpool_bf16(_ptr(dx), 2, H, W, 32, k, stride, pad, pad, pad_zero, Ho, Wo, _ptr(out), _stream()),
          "sa_maxpool_bf16")
    xp = F.pad(_bf(x).permute(0, 3, 1, 2), (pad, pad, pad, pad), value=0.0 if pad_zero else float("-inf"))
    ref = F.max_pool2d(xp, k, stride).permute(0, 2, 3, 1)
    assert torch.equal(out.float().cpu(), ref)


@pytest.mark.parametrize("pad,pad_zero,H,W,B", [(1, 1, 64, 96, 6), (1, 1, 37, 71, 3), (1, 0, 40, 130, 2), (0, 0, 33, 65, 4), (1, 1, 512, 512, 8)])
def test_maxpool3x3s2_tiled_plane_kernel_vs_torch(pad, pad_zero, H, W, B):
    """Round 5: MaxPooling2D(3, s2) on 16-channel planes (what ResNet's pool after the stem is, per plane) runs an LDS-tiled
    kernel (`maxpool3x3s2_c16_kernel`): every size class -- tiles inside, ragged right / bottom edges, odd sizes, both padding
    semantics (ZeroPadding2D zeros take part / Keras `same` taps are skipped) and the benchmark's 512 x 512 -- bit for bit
    torch's max pool of the same stored values."""
    from sleap_amd import _lib, ops
    from sleap_amd._lib import check
    from sleap_amd.ops import _ptr, _stream

    g = torch.Generator(device="cpu").manual_seed(H + W)
    x = torch.randn((B, H, W, 16), generator=g) - (1.0 if pad_zero else 0.0)
    dx = ops.to_bf16_padded(x.cuda())
    Ho, Wo = (H + 2 * pad - 3) // 2 + 1, (W + 2 * pad - 3) // 2 + 1
    out = torch.full((B, Ho, Wo, 16), float("nan"), dtype=TD, device="cuda")
    check(_lib.lib().sa_maxpool_bf16(_ptr(dx), B, H, W, 16, 3, 2, pad, pad, pad_zero, Ho, Wo, _ptr(out), _stream()), "sa_maxpool_bf16")
    xp = F.pad(_bf(x).permute(0, 3, 1, 2), (pad, pad, pad, pad), value=0.0 if pad_zero else float("-inf"))
    ref = F.max_pool2d(xp, 3, 2).permute(0, 2, 3, 1)
    assert torch.equal(out.float().cpu(), ref)


@pytest.mark.parametrize("B,H,W,Cin,Cout,ksize,affine,relu", [
    (2, 40, 56, 16, 16, 7, False, True),     # stem0_conv1 of a UNet stem (unet.py:105-127): k7, 16 -> 16
    (1, 33, 47, 32, 32, 7, False, True),     # odd sizes: every border case of the hardware zero fill
    (1, 24, 24, 24, 40, 5, True, True),      # channel padding on both sides, k5, BatchNormalization after the ReLU
    (1, 20, 28, 64, 128, 9, False, False),   # the largest window, two 64-channel cout tiles, linear
    (1, 18, 22, 16, 16, 4, False, True),     # even window: TF SAME pads (k - 1) // 2 before, k // 2 after
])
def test_convk_window_conv_vs_torch(B, H, W, Cin, Cout, ksize, affine, relu):
    """Conv2D(k x k, stride 1, same) on a feature tensor through the tap GEMM (sa_convk_bf16) vs fp32 torch on the same
    16-bit-rounded operands."""
    from sleap_amd import _lib, ops
    from sleap_amd._lib import check
    from sleap_amd.ops import _ptr, _stream

    g = torch.Generator(device="cpu").manual_seed(H + Cin + Cout + ksize)
    k = torch.randn((ksize, ksize, Cin, Cout), generator=g) * (2.0 / (Cin * ksize * ksize)) ** 0.5
    bias = 0.1 * torch.randn((Cout,), generator=g)
    x = torch.randn((B, H, W, Cin), generator=g)
    scale = 1.0 + 0.3 * torch.randn((Cout,), generator=g)
    shift = 0.2 * torch.randn((Cout,), generator=g)
    pb, pa = (ksize - 1) // 2, ksize // 2
    xp = F.pad(_bf(x).permute(0, 3, 1, 2), (pb, pa, pb, pa))
    y = F.conv2d(xp, _bf(k).permute(3, 2, 0, 1), bias).permute(0, 2, 3, 1)
    if relu:
        y = torch.relu(y)
    if affine:
        y = y * scale + shift
    cinp, coutp = ops.pad16(Cin), ops.pad16(Cout)
    dx = ops.to_bf16_padded(x.cuda().contiguous())
    pw = _pack_taps(k.reshape(ksize * ksize, Cin, Cout), cinp, coutp)
    bp = _padded(bias, coutp)
    ps = _padded(scale, coutp, 1.0) if affine else None
    pt = _padded(shift, coutp) if affine else None
    out = torch.full((B, H, W, coutp), 7.0, dtype=TD, device="cuda")
    check(_lib.lib().sa_convk_bf16(_ptr(dx), cinp, _ptr(pw), ksize, _ptr(bp), coutp, int(relu), B, H, W, _ptr(ps), _ptr(pt), None, 0,
                                   _ptr(out), _stream()), "sa_convk_bf16")
    got = ops.from_bf16(out, Cout).cpu()
    lim = 1e-2 * float(y.abs().max())
    assert float((got - y).abs().max()) <= lim, (float((got - y).abs().max()), lim)
    if coutp > Cout:
        assert float(out.float()[..., Cout:].abs().max()) == 0.0


def test_unet_with_stem_blocks_vs_oracle():
    """UNet.from_config with stem_stride=4 (unet.py:105-127, 250-278): two stem blocks of 7 x 7 convs (the first on the uint8
    image on the matrix cores, the others through the tap GEMM), pooled stem output as encoder input AND as the skip source at
    stride 4, bilinear decoder back to stride 4, a confidence-map head -- vs the fp32 oracle and the 16-bit-rounding oracle."""
    from sleap_amd.nn.architectures import build_unet_model_config, he_normal_weights

    cfg, shapes = build_unet_model_config((128, 160, 1), filters=8, filters_rate=2, max_stride=32, output_stride=4,
                                          stem_stride=4, heads=[("MultiInstanceConfmapsHead", 5, 4)])
    w = he_normal_weights(shapes, seed=5)
    rng = np.random.default_rng(5)
    net, res = _parity(cfg, w, rng.integers(0, 256, (2, 128, 160, 1), dtype=np.uint8), 3e-2, 2e-2)
    kinds = [op[0] for op in net.plan]
    assert kinds.count("imgconv") == 1 and sum(1 for op in net.plan if op[0] == "conv1x1" and op.ksize == 7) == 3, kinds
    print("UNet with stem blocks:", res)


def test_stacked_unet_with_stem_features_vs_oracle():
    """The reference's own stacked-UNet-with-stem architecture test shape (tests/nn/architectures/test_unet.py:159-196) at a
    smaller width: 2 stem blocks (k3 there), 2 stacks, both stack outputs."""
    from sleap_amd.nn.architectures import build_unet_model_config, he_normal_weights

    cfg, shapes = build_unet_model_config((96, 96, 1), filters=8, filters_rate=2, middle_block=True, up_interpolate=True,
                                          stacks=2, stem_blocks=2, down_blocks=3, up_blocks=3, stem_kernel_size=3)
    w = he_normal_weights(shapes, seed=6)
    rng = np.random.default_rng(6)
    net, _ = _parity(cfg, w, rng.integers(0, 256, (1, 96, 96, 1), dtype=np.uint8), 4e-2, 2e-2)
    assert len(net.outputs) == 2


# ------------------------------------------------------------------------------------------------
# ResNet backbones + UpsamplingStack (SURVEY.md §8a row a2'')
# ------------------------------------------------------------------------------------------------
def _resnet(h, w, cin=1, seed=3, residual_scale=0.25, **kw):
    """residual_scale 0.25: a random-init ResNet whose activations stay in the range of a trained one (and of fp16 storage);
    1.0 = plain He init, which leaves fp16's range at depth (test_resnet50_plain_he_init_...)."""
    from sleap_amd.nn.architectures import build_resnet_model_config, he_normal_weights

    cfg, shapes = build_resnet_model_config((h, w, cin), **kw)
    return cfg, he_normal_weights(shapes, seed=seed, residual_scale=residual_scale)


def test_resnet50_pretrained_style_transposed_concat_vs_oracle():
    """ResNet-50 with the input Lambdas (tile_channels + imagenet_preproc_v1), stride-32 features, transposed-conv
    (k4, BN) upsampling stack with concatenated skips to stride 4, heads at strides 4 and 8."""
    cfg, w = _resnet(128, 96, features_output_stride=32, pretrained=True,
                     upsampling=dict(output_stride=4, method="transposed_conv", skip_connections="concatenate"),
                     heads=[("MultiInstanceConfmapsHead", 5, 4), ("PartAffinityFieldsHead", 8, 8)])
    rng = np.random.default_rng(0)
    net, res = _parity(cfg, w, rng.integers(0, 256, (2, 128, 96, 1), dtype=np.uint8), 5e-2, 3e-2)
    kinds = [op[0] for op in net.plan]
    # 53 convs of the backbone: 36 are 1x1 (stride 1 / 2) launches of their own; round 4 folds the first stage's three 3x3 convs,
    # their three expand convs and two reduce convs of the following blocks into three bottleneck-tail launches
    assert "add" not in kinds and kinds.count("conv1x1") + 5 * (kinds.count("bneck") == 3) == 36 and kinds.count("convt2") == 3


def test_resnet50_plain_he_init_overflows_fp16_and_is_rescaled():
    """Plain He init (residual_scale 1): activations pass 65504 around conv4_block3. With range scaling off the fp16 build
    must say so on the first forward (engine range check -- the heads themselves can come out finite because ReLU scrubs
    the NaNs); the bf16 build matches the oracle at bf16's precision; the DEFAULT fp16 network folds power-of-two activation
    scales into its weights on that first forward (nn/range_scaling.py) and matches the oracle at fp16's precision -- the
    one storage mode that is both range safe and inside north_star's tolerance (tests/test_gpu_config_parity.py holds the
    end-to-end case)."""
    from sleap_amd import _lib
    from sleap_amd.nn.engine import DeviceNetwork

    cfg, w = _resnet(128, 96, residual_scale=1.0, features_output_stride=32, pretrained=True,
                     upsampling=dict(output_stride=4, method="transposed_conv", skip_connections="concatenate"),
                     heads=[("MultiInstanceConfmapsHead", 5, 4), ("PartAffinityFieldsHead", 8, 8)])
    x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (2, 128, 96, 1), dtype=np.uint8)).cuda()
    with pytest.raises(FloatingPointError, match="bf16"):
        DeviceNetwork(cfg, w, dtype="fp16", range_safe=False).forward(x)
    from oracle.keras_graph import KerasGraph, ensure_float

    ref = KerasGraph(cfg, w)(ensure_float(x.cpu().numpy()))
    err = {}
    for dt in ("bf16", "fp16"):
        net = DeviceNetwork(cfg, w, dtype=dt)
        outs = [o.cpu().numpy() for o in net.forward(x)]
        for o in outs:
            assert np.isfinite(o).all()
        err[dt] = max(float(np.abs(o - r).max() / np.abs(r).max()) for o, r in zip(outs, ref))
    ks = net.range_log2_scale
    assert ks is not None and min(ks.values()) <= -5, "the fp16 plan was not rescaled"
    # every stored tensor of the rescaled plan is far inside the format's range
    bufs = next(iter(net._buffers.values()))
    from sleap_amd.nn.range_scaling import TRIGGER

    assert max(float(t.abs().max()) for t in bufs.values() if t.dtype == torch.float16) <= TRIGGER * 1.01
    print("relative head error vs the fp32 oracle:", err)
    assert err["bf16"] <= 5e-2 and err["fp16"] <= 1e-2 and err["fp16"] < err["bf16"]
    assert _lib.DEFAULT_DTYPE in _lib.DTYPES


def test_tensor_absmax_kernel_vs_torch():
    """`sa_tensor_absmax` (the range scan's reduction; replaces torch.aminmax in the product path): largest finite magnitude
    and inf / NaN flags, both storage types + float32, sizes with a tail, values placed in the tail and in the body."""
    import ctypes as C

    from sleap_amd import _lib

    rng = np.random.default_rng(3)
    for dt, tdt in (("fp16", torch.float16), ("bf16", torch.bfloat16), ("f32", torch.float32)):
        h = _lib.lib("bf16" if dt == "f32" else dt)
        for n in (8, 1000, 4099, 1 << 20, (1 << 22) + 5):
            x = torch.from_numpy(rng.normal(0, 30, n).astype(np.float32)).cuda().to(tdt)
            x[rng.integers(n)] = -3000.0
            x[n - 1] = 2000.0
            for special in (None, float("inf"), float("-inf"), float("nan")):
                y = x.clone()
                if special is not None:
                    y[rng.integers(n)] = special
                out = torch.full((2,), 7.0, device="cuda")
                _lib.check(h.sa_tensor_absmax(C.c_void_p(y.data_ptr()), n, 1 if dt == "f32" else 0, C.c_void_p(out.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), "sa_tensor_absmax")
                m = float(out[0])
                flags = int(out[1:].cpu().numpy().view(np.uint32)[0])
                fin = y[torch.isfinite(y)]
                assert m == float(fin.abs().max()), (dt, n, special, m)
                want = 0 if special is None else (2 if special != special else 1)
                assert flags == want, (dt, n, special, flags)


def test_range_calibration_is_explicit_persistable_and_keeps_the_master_weights():
    """range scales must not depend on which batch happens to come first. `calibrate_range(frames)` measures on ALL
    given frames and folds once; the exponents can be persisted and handed to a new network (`range_log2_scale=`), which then
    computes the same bits with no first-batch measurement; `master_weights` stay the model's float32 weights."""
    from sleap_amd.nn.engine import DeviceNetwork

    cfg, w = _resnet(128, 96, residual_scale=1.0, features_output_stride=32, pretrained=True,
                     upsampling=dict(output_stride=4, method="transposed_conv", skip_connections="concatenate"),
                     heads=[("MultiInstanceConfmapsHead", 5, 4), ("PartAffinityFieldsHead", 8, 8)])
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.integers(0, 256, (11, 128, 96, 1), dtype=np.uint8)).cuda()  # 11 frames: chunks of 8 + an overlap
    net = DeviceNetwork(cfg, w, dtype="fp16")
    ks = net.calibrate_range(x)
    assert ks and min(ks.values()) <= -5 and net.range_log2_scale == ks
    assert all(np.array_equal(net.master_weights[k], w[k]) for k in w) and net.master_weights is w
    assert any(not np.array_equal(net.weights[k], w[k]) for k in w)  # the compiled weights carry the fold
    a = [o.clone() for o in net.forward(x[:2].contiguous())]
    import json

    net2 = DeviceNetwork(cfg, w, dtype="fp16", range_log2_scale=json.loads(json.dumps(ks)))
    b = net2.forward(x[:2].contiguous())
    assert net2.range_log2_scale == ks
    for p, q in zip(a, b):
        assert torch.equal(p, q)
    # a network that fits is left alone: no exponents, same weights object
    cfg3, w3 = _resnet(64, 64, residual_scale=0.25, features_output_stride=32)
    net3 = DeviceNetwork(cfg3, w3, dtype="fp16")
    assert net3.calibrate_range(x[:2, :64, :64].contiguous()) == {} and net3.weights is w3


def test_finite_but_unscalable_range_is_a_warning_not_an_error():
    """ADVICE r3 (engine.py:1252): a finite first-batch maximum within 4x of the range whose tensor cannot be rescaled (here: the
    model OUTPUT of a one-conv network) used to raise 'activations left the range'; it overflowed nothing -> a warning."""
    from sleap_amd.nn import architectures as A
    from sleap_amd.nn.engine import DeviceNetwork

    layers = [{"class_name": "InputLayer", "name": "input", "config": {"batch_input_shape": [None, 32, 32, 1]}, "inbound_nodes": []},
              {"class_name": "Conv2D", "name": "c0", "inbound_nodes": [[["input", 0, 0, {}]]],
               "config": {"filters": 16, "kernel_size": [3, 3], "strides": [1, 1], "padding": "same", "activation": "relu", "use_bias": True}},
              {"class_name": "Conv2D", "name": "c1", "inbound_nodes": [[["c0", 0, 0, {}]]],
               "config": {"filters": 16, "kernel_size": [3, 3], "strides": [1, 1], "padding": "same", "activation": "relu", "use_bias": True}}]
    cfg = {"class_name": "Functional", "config": {"layers": layers, "input_layers": [["input", 0, 0]], "output_layers": [["c1", 0, 0]]}}
    w = {"c0/kernel": np.full((3, 3, 1, 16), 0.5, np.float32), "c0/bias": np.zeros(16, np.float32),
         "c1/kernel": np.full((3, 3, 16, 16), 40.0, np.float32), "c1/bias": np.zeros(16, np.float32)}
    x = torch.full((1, 32, 32, 1), 255, dtype=torch.uint8).cuda()
    net = DeviceNetwork(cfg, w, dtype="fp16", fuse_stem=False)
    with pytest.warns(UserWarning, match="within 4x"):
        out = net.forward(x)[0]
    assert torch.isfinite(out).all() and 65504 / 4 < float(out.max()) < 65504 and net.range_log2_scale is None


def test_resnet50_bottleneck_tails_fused_vs_unfused_and_oracle():
    """Round 4: the three bottleneck tails of ResNet's first stage (64-map 3x3 conv -> 1x1 expand + BN + shortcut Add + ReLU ->
    the next block's 1x1 reduce) run as ONE launch each (sa_conv3x3_bneck_bf16). Same operations per value as the un-fused
    launches (the matrix cores add a k-step's products in another order): every model output within 2 fp16 ulp of the output's
    range of the un-fused plan, and within the usual tolerance of the fp32 oracle; both layouts."""
    from oracle.keras_graph import KerasGraph, ensure_float
    from sleap_amd.nn.engine import DeviceNetwork

    cfg, w = _resnet(128, 160, features_output_stride=32, pretrained=True,
                     upsampling=dict(output_stride=4, method="transposed_conv", skip_connections="concatenate"),
                     heads=[("MultiInstanceConfmapsHead", 5, 4), ("PartAffinityFieldsHead", 8, 8)])
    x = torch.from_numpy(np.random.default_rng(7).integers(0, 256, (3, 128, 160, 1), dtype=np.uint8)).cuda()
    ref = KerasGraph(cfg, w)(ensure_float(x.cpu().numpy()))
    for layout in (None, "nhwc"):
        fused = DeviceNetwork(cfg, w, dtype="fp16", layout=layout)
        plain = DeviceNetwork(cfg, w, dtype="fp16", layout=layout, fuse_bneck=False)
        kinds = [op[0] for op in fused.plan]
        assert kinds.count("bneck") == 3 and len(fused.plan) == len(plain.plan) - 5  # 3 x (C, X) + 2 x Y fewer launches
        assert [op[3] is not None for op in fused.plan if op[0] == "bneck"] == [True, True, False]
        a = [o.cpu().numpy() for o in fused.forward(x)]
        b = [o.cpu().numpy() for o in plain.forward(x)]
        for u, v, r in zip(a, b, ref):
            rng = float(np.abs(r).max())
            assert np.isfinite(u).all()
            assert float(np.abs(u - v).max()) <= 2 * 2.0 ** -10 * rng, (layout, float(np.abs(u - v).max()), rng)
            assert float(np.abs(u - r).max()) <= 3e-2 * rng


def test_resnet50_decoder_heads_fused_behind_batchnorm_vs_unfused_and_oracle(monkeypatch):
    """Round 4: the decoder's Conv + BatchNormalization + ReLU in front of a head takes the head into its epilogue
    (sa_conv3x3_ex_heads_bf16: the heads see the value the extended epilogue stores). Fused vs the stand-alone head launches
    (SA_FUSE_EXT_HEADS=0) vs the fp32 oracle, on planes and on NHWC; and the per-launch Python loop (bench.py --layers,
    tools/net_profile.py) is the plan's launch sequence bit for bit -- it passes the layout to every launch."""
    from oracle.keras_graph import KerasGraph, ensure_float
    from sleap_amd.nn.engine import DeviceNetwork

    cfg, w = _resnet(128, 160, features_output_stride=32, pretrained=True,
                     upsampling=dict(output_stride=4, method="transposed_conv", skip_connections="concatenate", filters=64, refine_convs=2),
                     heads=[("MultiInstanceConfmapsHead", 24, 4), ("PartAffinityFieldsHead", 46, 8)])  # configs[4]'s heads
    x = torch.from_numpy(np.random.default_rng(11).integers(0, 256, (3, 128, 160, 1), dtype=np.uint8)).cuda()
    ref = KerasGraph(cfg, w)(ensure_float(x.cpu().numpy()))
    for layout in (None, "nhwc"):
        fused = DeviceNetwork(cfg, w, dtype="fp16", layout=layout)
        monkeypatch.setenv("SA_FUSE_EXT_HEADS", "0")
        plain = DeviceNetwork(cfg, w, dtype="fp16", layout=layout)
        monkeypatch.delenv("SA_FUSE_EXT_HEADS")
        assert [op[0] for op in fused.plan].count("head") == 0 and [op[0] for op in plain.plan].count("head") == 2
        assert sum(1 for op in fused.plan if op[0] == "conv" and op.heads and op.ext is not None) == 2
        a = [o.clone() for o in fused.forward(x)]
        b = [o.clone() for o in plain.forward(x)]
        prof = []
        c = [o.clone() for o in fused.forward(x, profile=prof)]
        assert len(prof) == len(fused.plan)
        for u, v, q, r in zip(a, b, c, ref):
            assert torch.equal(u, q), layout
            u, v = u.cpu().numpy(), v.cpu().numpy()
            rng = float(np.abs(r).max())
            assert np.isfinite(u).all()
            # same stored features (bitwise the un-fused tensor); the head GEMM: fp32 FMA chain vs matrix cores on hi + lo weights
            assert float(np.abs(u - v).max()) <= 2e-3 * rng, (layout, float(np.abs(u - v).max()), rng)
            assert float(np.abs(u - r).max()) <= 3e-2 * rng


def test_resnet50_stride16_bilinear_add_vs_oracle():
    """features_output_stride 16 (conv5 unstrided, dilated 1x1 convs = no-op), bilinear upsampling with additive
    skips through 1x1 projections; RGB input without the ImageNet Lambdas."""
    cfg, w = _resnet(64, 96, cin=3, features_output_stride=16, pretrained=False,
                     upsampling=dict(output_stride=4, method="interpolation", skip_connections="add", filters=32),
                     heads=[("MultiInstanceConfmapsHead", 5, 4)])
    rng = np.random.default_rng(1)
    net, _ = _parity(cfg, w, rng.integers(0, 256, (1, 64, 96, 3), dtype=np.uint8), 5e-2, 3e-2)
    assert "add" not in [op[0] for op in net.plan]


def test_resnet50_backbone_features_vs_oracle():
    """Backbone only (what tests/nn/architectures/test_resnet.py builds): (B, H/32, W/32, 2048) features."""
    cfg, w = _resnet(96, 96, features_output_stride=32)
    rng = np.random.default_rng(2)
    net, _ = _parity(cfg, w, rng.integers(0, 256, (2, 96, 96, 1), dtype=np.uint8), 5e-2, 3e-2)
    assert net.outputs[0].c == 2048


@pytest.mark.parametrize("B,H,W,Cin,CinW,Cout,stride,mean,relu,affine", [
    (2, 64, 96, 1, 1, 128, 2, False, True, True),    # hourglass stem (pad 2/3 from TF SAME)
    (1, 70, 50, 3, 3, 64, 2, True, True, False),     # ResNet stem on RGB with the ImageNet means, borders everywhere
    (2, 128, 128, 1, 3, 64, 2, True, True, False),   # tile_channels: grayscale image, 3 weight channels; interior tiles
    (1, 40, 72, 1, 1, 24, 1, False, False, True),    # stride 1, channel padding (24 -> 32)
    (1, 33, 47, 3, 3, 16, 2, False, True, False),    # odd sizes
])
def test_imgconv_mfma_vs_torch(B, H, W, Cin, CinW, Cout, stride, mean, relu, affine):
    """k7 stem conv on the matrix cores vs fp32 torch on the same uint8 image: hi+lo bf16 weights (16 mantissa bits), exact
    pixel operands, fp32 accumulation -> only the final bf16 rounding (2^-9) plus ~2^-16 weight error remain."""
    import ctypes as C

    from sleap_amd import _lib, ops
    from sleap_amd._lib import check
    from sleap_amd.ops import _ptr, _stream

    g = torch.Generator(device="cpu").manual_seed(H + W + Cout)
    img = torch.randint(0, 256, (B, H, W, Cin), generator=g, dtype=torch.uint8)
    kern = torch.randn((7, 7, CinW, Cout), generator=g) * (2.0 / (49 * CinW)) ** 0.5 * (0.02 if mean else 1.0)
    bias = 0.1 * torch.randn((Cout,), generator=g)
    scale = 1.0 + 0.3 * torch.randn((Cout,), generator=g)
    shift = 0.2 * torch.randn((Cout,), generator=g)
    means = torch.tensor([123.68, 116.779, 103.939])[:CinW]
    x = img.float()
    if Cin == 1 and CinW == 3:
        x = x.repeat(1, 1, 1, 3)
    if mean:
        x = x - means  # (u8 / 255) * 255 - mean, already in weight-channel order
    else:
        x = x * (1.0 / 255.0)
    if mean:  # ZeroPadding2D(3) + valid conv (resnet.py:109-114)
        pt = pl = 3
        Ho, Wo = (H + 6 - 7) // stride + 1, (W + 6 - 7) // stride + 1
        xp = F.pad(x.permute(0, 3, 1, 2), (3, 3, 3, 3))
    else:  # TF SAME
        Ho, Wo = -(-H // stride), -(-W // stride)
        ph, pw = max((Ho - 1) * stride + 7 - H, 0), max((Wo - 1) * stride + 7 - W, 0)
        pt, pl = ph // 2, pw // 2
        xp = F.pad(x.permute(0, 3, 1, 2), (pl, pw - pl, pt, ph - pt))
    y = F.conv2d(xp.double(), kern.permute(3, 2, 0, 1).double(), bias.double(), stride=stride).float().permute(0, 2, 3, 1)
    assert y.shape[1:3] == (Ho, Wo)
    if relu:
        y = torch.relu(y)
    if affine:
        y = y * scale + shift
    h = _lib.lib()
    coutp = ops.pad16(Cout)
    packed = np.zeros((h.sa_imgconv_packed_elems(7, CinW, coutp),), np.uint16)
    bias_io = np.zeros((coutp,), np.float32)
    bias_io[:Cout] = bias.numpy()
    wk = np.ascontiguousarray(kern.numpy(), dtype=np.float32)
    sc = np.full((CinW,), 1.0 if mean else 1.0 / 255.0, np.float32)
    mn = np.ascontiguousarray(means.numpy(), dtype=np.float32) if mean else None
    vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
    check(h.sa_imgconv_pack(vp(wk), 7, CinW, Cout, coutp, vp(sc), vp(mn), vp(packed), vp(bias_io)), "sa_imgconv_pack")
    dw = torch.from_numpy(packed.view(np.int16)).cuda()
    db = torch.from_numpy(bias_io).cuda()
    ps = _padded(scale, coutp, 1.0) if affine else None
    psh = _padded(shift, coutp) if affine else None
    out = torch.full((B, Ho, Wo, coutp), 7.0, dtype=TD, device="cuda")
    dimg = img.cuda().contiguous()
    check(h.sa_imgconv_u8_bf16(_ptr(dimg), B, H, W, Cin, CinW, 7, stride, pt, pl, Ho, Wo, _ptr(dw), _ptr(db), coutp, int(relu),
                               int(mean), _ptr(ps), _ptr(psh), _ptr(out), _stream()), "sa_imgconv_u8_bf16")
    got = ops.from_bf16(out, Cout).cpu()
    lim = 2.0 ** -8 * float(y.abs().max())
    assert float((got - y).abs().max()) <= lim, (float((got - y).abs().max()), lim)
    if coutp > Cout:
        assert float(out.float()[..., Cout:].abs().max()) == 0.0
