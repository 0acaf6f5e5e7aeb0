"""Where does bottomup_postproc_kernel (one workgroup per frame) spend its cycles? Stage sums of the instrumented build:

    python tools/build_alt.py libsleap_amd_fp16_ppstamp.so postproc.hip -DSA_POSTPROC_STAMP=1
    SLEAP_AMD_LIB_FP16=sleap_amd/lib/libsleap_amd_fp16_ppstamp.so python tools/pp_stamp_probe.py

The kernel runs ALONE here (every call synchronised); s_memtime ticks are shader cycles."""
import sys, ctypes as C, torch
sys.path.insert(0, ".")
from sleap_amd import _lib
from sleap_amd.benchmark_model import build_benchmark_predictor
from sleap_amd.synth import render_flies
h = _lib.lib("fp16")
h.sa_pp_stamp_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
names = ["sort + refine", "bucket + score", "group fill", "match (wave per edge)", "group (wave 0)"]
for B in (8, 64):
    pred = build_benchmark_predictor(1024, 1024, batch_size=B, seed=0)[0]
    layer = pred.inference_model.bottomup_layer
    layer.assume_inputs_ready = True
    fr = torch.from_numpy(render_flies(B, 1024, 1024, n_animals=4, seed=100)[0]).cuda()
    for _ in range(5):
        pred.inference_model.call(fr); torch.cuda.synchronize()
    st = (C.c_ulonglong * 8)(); h.sa_pp_stamp_read(st, 1)
    for _ in range(20):
        pred.inference_model.call(fr); torch.cuda.synchronize()   # synchronised: the kernel runs ALONE
    h.sa_pp_stamp_read(st, 1)
    n = st[5]
    tot = sum(st[i] for i in range(5)) / n
    print(f"batch {B}: {n} workgroups; shader cycles per workgroup: total {tot:.0f}")
    for i in range(5):
        print(f"   {names[i]:24s} {st[i] / n:9.0f}  {st[i] / n / tot:5.1%}")
