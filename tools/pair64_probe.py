"""Where do the waves of convpair64_kernel (encoder block 2 in one launch) spend their cycles?

    python tools/build_alt.py libsleap_amd_fp16_p64stamp.so convpair64.hip -DSA_PAIR64_STAMP=1
    SLEAP_AMD_LIB_FP16=sleap_amd/lib/libsleap_amd_fp16_p64stamp.so python tools/pair64_probe.py [B]

Every wave of the instrumented build sums the shader cycles (s_memtime) of the segments of its tiles; printed per tile and wave:
the MFMA floor of a segment is 32 cycles x MFMAs x 2 waves per SIMD (A stage: 45, B stage: 36 MFMAs per wave)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from sleap_amd import _lib, ops
from sleap_amd._lib import check
from sleap_amd.ops import _ptr, _stream

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = W = 256
h = _lib.lib("fp16")
stamped = hasattr(h, "sa_pair64_stamp_reset")
g = torch.Generator().manual_seed(0)
ka = (torch.randn((3, 3, 32, 64), generator=g) * (2.0 / (9 * 32)) ** 0.5).numpy()
kb = (torch.randn((3, 3, 64, 64), generator=g) * (2.0 / (9 * 64)) ** 0.5).numpy()
wa, wb = ops.pack_conv3x3_weights(ka, 32, dtype="fp16"), ops.pack_conv3x3_weights(kb, 64, dtype="fp16")
ba, bb = torch.zeros((64,), device="cuda"), torch.zeros((64,), device="cuda")
x = torch.randn((B, H, W, 32), device="cuda").clamp_(min=0).to(torch.float16)
out = torch.empty((B, H, W, 64), dtype=torch.float16, device="cuda")
outp = torch.empty((B, H // 2, W // 2, 64), dtype=torch.float16, device="cuda")


def run():
    check(h.sa_conv3x3_pair_bf16(_ptr(x), 32, _ptr(wa), _ptr(ba), 1, 64, _ptr(wb), _ptr(bb), 1, 64, B, H, W, _ptr(out), _ptr(outp),
                                 _lib.LAYOUT_PLANES16, _stream()), "pair")


for _ in range(3):
    run()
torch.cuda.synchronize()
reps = 10
if stamped:
    h.sa_pair64_stamp_reset()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * B * H * W * 9 * (32 * 64 + 64 * 64)
print(f"convpair64 {B} frames of {H}x{W}: {ms:.4f} ms per launch, {fl / ms / 1e9:.0f} TFLOP/s (algorithmic)")
if stamped:
    h.sa_pair64_stamp_read.argtypes = [C.POINTER(C.c_ulonglong)]
    st = (C.c_ulonglong * 24)()
    h.sa_pair64_stamp_read(st)
    v = [int(t) for t in st]
    waves, tiles = v[0], v[1]
    names = {15: "tile set-up (bias -> accumulators)", 8: "barrier before A0", 3: "A0 + A1 bodies (floor 2 x 2880)", 9: "barrier before A1",
             4: "epilogue a", 10: "barrier before B0", 11: "... B1", 12: "... B2", 13: "... B3", 5: "B0..B3 bodies (floor 4 x 2304)",
             14: "wait before epilogue b", 6: "epilogue b"}
    print(f"waves {waves // reps} per launch, tiles per wave {tiles / waves:.1f}; cycles per tile and wave (life / tiles = {v[2] / tiles:.0f}):")
    for i in (15, 8, 3, 9, 4, 10, 5, 11, 12, 13, 14, 6):
        extra = f"   (memory wait alone {v[i + 8] / tiles:6.0f})" if 8 <= i <= 13 else ""
        print(f"  {names[i]:42s} {v[i] / tiles:8.0f}{extra}")
