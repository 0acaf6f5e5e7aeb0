"""Builds libsleap_amd.so (HIP kernels + C ABI) for gfx950 with plain hipcc, in-tree.

    python -m sleap_amd.build [--force]

hipcc cross-compiles without a GPU. The shared object lands in `sleap_amd/lib/` (git-ignored,
travels with the source snapshot to the GPU box).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsleap_amd.so")
ARCH = "gfx950"

# (source, extra flags). postproc.hip must evaluate fp32 op-by-op (no FMA contraction).
SOURCES = [
    ("postproc.hip", ["-ffp-contract=off"]),
    ("layers.hip", []),
    # the MFMA kernels never see NaNs they must preserve: without this every fmaxf carries two canonicalising v_max
    ("conv3x3.hip", ["-fno-honor-nans"]),
    ("stem16.hip", ["-fno-honor-nans"]),
    ("tapconv.hip", ["-fno-honor-nans"]),
    ("convpair.hip", ["-fno-honor-nans"]),
    ("imgconv.hip", ["-fno-honor-nans", "-std=c++20"]),
    ("tracker.hip", []),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "sleap_amd.h"))
    return hdrs


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    rebuilt = False
    dep_mtime = max(os.path.getmtime(h) for h in _deps())
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), dep_mtime):
            cmd = [hipcc, "-c", s, "-o", o] + COMMON + extra
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            rebuilt = True
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
