"""Micro-benchmark of one 3x3 conv shape with / without the extended epilogue (BatchNormalization affine, residual), NHWC or planes.
usage: conv_ext_bench.py Cin Cout H B [planes=1] [reps=20] [input: dense|relu|zeros]   (env SA_CONV_MT4 etc. apply)

The input statistics matter: the same launch runs at different clocks on dense random values, on post-ReLU values (half
of them zero) and on zeros (profiles/r03_ab_session.md section 5)."""
import sys

import torch

sys.path.insert(0, ".")
from sleap_amd import _lib, ops
from sleap_amd.ops import _ptr, _stream, check

Cin, Cout, H, B = [int(a) for a in sys.argv[1:5]]
planes = int(sys.argv[5]) if len(sys.argv) > 5 else 1
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
g = torch.Generator().manual_seed(0)
k = (torch.randn((3, 3, Cin, Cout), generator=g) * (2.0 / (9 * Cin)) ** 0.5).numpy()
pw = ops.pack_conv3x3_weights(k, Cin, 0, "fp16")
bias = torch.zeros((Cout,), device="cuda")
ps, pt = torch.ones((Cout,), device="cuda"), torch.zeros((Cout,), device="cuda")
inp = sys.argv[7] if len(sys.argv) > 7 else "dense"
x = torch.randn((B, H, H, Cin), device="cuda")
x = (x.clamp_min(0) if inp == "relu" else x * 0 if inp == "zeros" else x).to(torch.float16)
res = torch.randn((B, H, H, Cout), device="cuda").to(torch.float16)
out = torch.empty((B, H, H, Cout), dtype=torch.float16, device="cuda")
mode = _lib.LAYOUT_PLANES16 if planes else 0
h = _lib.lib("fp16")
fl = 2.0 * B * H * H * Cin * Cout * 9


def run(kind):
    if kind == "plain":
        return h.sa_conv3x3_bf16(_ptr(x), Cin, None, 0, mode, _ptr(pw), _ptr(bias), Cout, 1, B, H, H, _ptr(out), None, _stream())
    return h.sa_conv3x3_ex_bf16(_ptr(x), Cin, None, 0, mode, _ptr(pw), _ptr(bias), Cout, 1, B, H, H, _ptr(out), None, _ptr(ps), _ptr(pt),
                                _ptr(res) if kind == "affine+res" else None, 0, 0, _stream())


for kind in ("plain", "affine", "affine+res"):
    for _ in range(3):
        check(run(kind), kind)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(kind)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"conv {Cin}->{Cout} @{H} B={B} planes={planes} {inp} {kind:11s}: {ms:.4f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
