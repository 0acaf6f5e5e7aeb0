"""Where do the waves of conv3x3_dma_kernel spend their cycles?  (stall attribution of the dominant kernel)

Runs the plain 3x3 layers of the benchmark plan (64 frames of 1024 x 1024 -> the layer shapes below), one launch shape at a time,
on the INSTRUMENTED build of csrc/conv3x3.hip (-DSA_CONV_STAMP=1, tools/build_alt.py: every wave sums the shader cycles of five
segments of its life with s_memtime and adds them to a device array at its end) and prints, per layer:

    time per launch (HIP events; the instrumented build -- and, with a second library, the product build beside it)
    share of the waves' summed life spent in: tile prologue | s_waitcnt vmcnt | barrier | chunk body | epilogue
    chunk-body cycles per chunk against the 36 x 32 = 1152 (MT = 2) / 18 x 32 = 576 (MT = 1) cycles of its MFMAs alone

        python tools/build_alt.py libsleap_amd_fp16_stamp.so conv3x3.hip -DSA_CONV_STAMP=1
        SLEAP_AMD_LIB_FP16=sleap_amd/lib/libsleap_amd_fp16_stamp.so python tools/stall_probe.py [B] > profiles/r05_stall_attribution.md

A wave's "chunk body" holds its own MFMAs (36 per chunk, 32 cycles each when the pipe is its alone) AND the time the SIMD's other
three waves keep the matrix pipe busy, so body / 1152 ~ 4 is a saturated pipe shared by four waves; what is lost shows up as
body / 1152 > 4 (issue stalls inside the body) or as the other four columns (nothing of this wave is issued there).
"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from sleap_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
# (C0, C1, Cout, H, pooled) of the plain conv3x3_dma launches of the benchmark plan, in plan order
LAYERS = [(32, 0, 64, 256, False), (64, 0, 64, 256, True), (64, 0, 128, 128, False), (128, 0, 128, 128, True), (128, 0, 256, 64, False),
          (256, 0, 256, 64, True), (256, 0, 512, 32, False), (512, 0, 512, 32, False), (256, 512, 256, 64, False), (256, 0, 256, 64, False),
          (128, 256, 128, 128, False)]
h = _lib.lib("fp16")
if not hasattr(h, "sa_conv3x3_stamp_reset"):
    raise SystemExit("this library was not built with -DSA_CONV_STAMP=1 (see the docstring)")
h.sa_conv3x3_stamp_reset.restype = C.c_int
h.sa_conv3x3_stamp_read.restype = C.c_int
h.sa_conv3x3_stamp_read.argtypes = [C.POINTER(C.c_ulonglong)]
print(f"# stall attribution of `conv3x3_dma_kernel` per layer ({B} frames, fp16 storage, 16-channel planes, random data)\n")
print("| layer | ms / launch | TFLOP/s | waves | chunks / wave | prologue | vmcnt wait | barrier | chunk body | epilogue | body cycles / chunk | / MFMA-only |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
g = torch.Generator().manual_seed(0)
for C0, C1, Cout, H, pooled in LAYERS:
    k = (torch.randn((3, 3, C0 + C1, Cout), generator=g) * (2.0 / (9 * (C0 + C1))) ** 0.5).numpy()
    pw = ops.pack_conv3x3_weights(k, C0, C1, dtype="fp16")
    coutp = ops.pad16(Cout)
    bias = torch.zeros((coutp,), device="cuda")
    x0 = torch.randn((B, H, H, C0), device="cuda").clamp_(min=0).to(torch.float16)
    x1 = torch.randn((B, H, H, C1), device="cuda").clamp_(min=0).to(torch.float16) if C1 else None
    mode = (1 if C1 else 0) | _lib.LAYOUT_PLANES16

    def run():
        return ops.conv3x3(x0, x1, mode, pw, bias, coutp, True, (H, H), full=True, pooled=pooled)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    reps = 10
    h.sa_conv3x3_stamp_reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    st = (C.c_ulonglong * 8)()
    h.sa_conv3x3_stamp_read(st)
    waves, pro, vm, bar, body, epi, life, chunks = [int(v) for v in st]
    fl = 2.0 * B * H * H * (C0 + C1) * Cout * 9
    mt = 2 if coutp >= 64 else 1
    tot = max(pro + vm + bar + body + epi, 1)
    per_chunk = body / max(chunks, 1)
    name = f"{C0}{'+' + str(C1) if C1 else ''}->{Cout} @{H}" + (" +pool" if pooled else "")
    print(f"| {name} | {ms:.4f} | {fl / ms / 1e9:.0f} | {waves // reps} | {chunks / max(waves, 1):.1f} | {pro / tot:.3f} | {vm / tot:.3f} | "
          f"{bar / tot:.3f} | {body / tot:.3f} | {epi / tot:.3f} | {per_chunk:.0f} | {per_chunk / (mt * 18 * 32):.2f} |")
    del x0, x1
print("\n(shares of the five segments' sum; `waves` per launch; life = sum over the waves of exit - entry, "
      "covered by the segments to within the stamps' own cost)")
