"""Micro-benchmark of sa_upsample2x_bf16 (bilinear) on the three shapes of the benchmark model. usage: up_bench.py [reps]"""
import sys

import torch

sys.path.insert(0, ".")
from sleap_amd import _lib
from sleap_amd._lib import check
from sleap_amd.ops import _ptr, _stream

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
h = _lib.lib()
for (H, C) in ((32, 512), (64, 256), (128, 128)):
    B = 64
    x = torch.randn((B, H, H, C), device="cuda").to(torch.bfloat16)
    o = torch.empty((B, 2 * H, 2 * H, C), dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        check(h.sa_upsample2x_bf16(_ptr(x), B, H, H, C, 1, _ptr(o), _stream()), "up")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        check(h.sa_upsample2x_bf16(_ptr(x), B, H, H, C, 1, _ptr(o), _stream()), "up")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gb = (x.numel() + o.numel()) * 2 / 1e9
    print(f"up {C}ch {H}->{2 * H}: {ms:.4f} ms  {gb / ms:.2f} TB/s")
