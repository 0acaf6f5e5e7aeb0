"""Micro-benchmark of one sa_conv3x3_bf16 shape (for rocprofv3 runs).
usage: conv_bench.py C0 C1 Cout H W B mode [reps] [pooled]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sleap_amd import ops

C0, C1, Cout, H, W, B, mode = [int(a) for a in sys.argv[1:8]]
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 20
pooled = bool(int(sys.argv[9])) if len(sys.argv) > 9 else False
g = torch.Generator().manual_seed(0)
k = (torch.randn((3, 3, C0 + C1, Cout), generator=g) * (2.0 / (9 * (C0 + C1))) ** 0.5).numpy()
pw = ops.pack_conv3x3_weights(k, C0, C1)
coutp = ops.pad16(Cout)
bias = torch.zeros((coutp,), device="cuda")
x0 = torch.randn((B, H, W, ops.pad16(C0)), device="cuda").to(torch.bfloat16)
x1 = torch.randn((B, H, W, ops.pad16(C1)), device="cuda").to(torch.bfloat16) if C1 else None
for _ in range(3):
    out = ops.conv3x3(x0, x1, mode, pw, bias, coutp, True, (H, W), full=not pooled, pooled=pooled)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    out = ops.conv3x3(x0, x1, mode, pw, bias, coutp, True, (H, W), full=not pooled, pooled=pooled)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * B * H * W * (C0 + C1) * Cout * 9
print(f"conv {C0}+{C1}->{Cout} @{H}x{W} B={B} mode={mode} pooled={pooled}: {ms:.4f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
