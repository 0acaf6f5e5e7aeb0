"""A/B of the activation layout for sa_conv3x3_bf16: NHWC vs 16-channel planes, per layer shape of the benchmark UNet.
Checks that both layouts give bitwise the same result, then times them alternately (HIP events on the current stream).
usage: python tools/planar_ab.py [B] [reps]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from sleap_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
DT = torch.float16

# (C0, C1, Cout, H, pooled_out, full_out)
SHAPES = [(32, 0, 64, 256, False, True), (64, 0, 64, 256, True, True), (64, 0, 128, 128, False, True),
          (128, 0, 128, 128, True, True), (128, 0, 256, 64, False, True), (256, 0, 256, 64, True, True),
          (256, 0, 512, 32, False, True), (512, 0, 512, 32, False, True), (256, 512, 256, 64, False, True),
          (256, 0, 256, 64, False, True), (128, 256, 128, 128, False, True), (64, 128, 64, 256, False, True)]


def run(layout, x0, x1, pw, bias, cout, H, full, pooled):
    mode = (1 if x1 is not None else 0) | (_lib.LAYOUT_PLANES16 if layout else 0)
    return ops.conv3x3(x0, x1, mode, pw, bias, cout, True, (H, H), full=full, pooled=pooled)


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


g = torch.Generator().manual_seed(0)
tot = [0.0, 0.0]
print("| layer | NHWC ms | planes ms | NHWC ms (2) | planes ms (2) | same bits |")
print("|---|---|---|---|---|---|")
for (C0, C1, Cout, H, pooled, full) in SHAPES:
    k = (torch.randn((3, 3, C0 + C1, Cout), generator=g) * (2.0 / (9 * (C0 + C1))) ** 0.5).numpy()
    pw = ops.pack_conv3x3_weights(k, C0, C1, dtype="fp16")
    bias = torch.zeros((Cout,), device="cuda")
    x0 = torch.relu(torch.randn((B, H, H, C0), generator=g)).to(DT).cuda()
    x1 = torch.relu(torch.randn((B, H, H, C1), generator=g)).to(DT).cuda() if C1 else None
    p0, p1 = ops.to_planes16(x0), (ops.to_planes16(x1) if C1 else None)
    a = run(0, x0, x1, pw, bias, Cout, H, full, pooled)
    b = run(1, p0, p1, pw, bias, Cout, H, full, pooled)
    a = a if isinstance(a, tuple) else (a,)
    b = b if isinstance(b, tuple) else (b,)
    same = all(torch.equal(u, ops.from_planes16(v)) for u, v in zip(a, b))
    t = []
    for _ in range(2):
        t.append(timeit(lambda: run(0, x0, x1, pw, bias, Cout, H, full, pooled)))
        t.append(timeit(lambda: run(1, p0, p1, pw, bias, Cout, H, full, pooled)))
    tot[0] += min(t[0], t[2])
    tot[1] += min(t[1], t[3])
    print(f"| {C0}+{C1}->{Cout} @{H}{' +pool' if pooled else ''} | {t[0]:.4f} | {t[1]:.4f} | {t[2]:.4f} | {t[3]:.4f} | {same} |", flush=True)
print(f"| sum of minima | {tot[0]:.3f} | {tot[1]:.3f} | | | |")
