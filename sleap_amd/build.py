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
    # + VGPR-form MFMA: left to itself the compiler puts these kernels' accumulators in AGPRs and pays one v_accvgpr_read
    # per accumulator register in the (VALU-bound) epilogues: stem16 0.840 -> 0.804 ms, imgconv 0.210 -> 0.137 ms (A/B on one
    # box, tools/ab_lib.sh). The 8-wave conv3x3 kernels are VGPR-form already; forcing it on the others changed nothing.
    ("stem16.hip", ["-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form"]),
    # (round 4: VGPR-form here too -- 64 accumulators in AGPRs cost 192 v_accvgpr moves per tile and 184-220 registers in
    #  total = two waves per SIMD; in VGPRs 124-156 = three)
    ("tapconv.hip", ["-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form"]),
    ("convpair.hip", ["-fno-honor-nans"]),
    ("convpair64.hip", ["-fno-honor-nans"]),
    ("imgconv.hip", ["-fno-honor-nans", "-std=c++20", "-mllvm", "-amdgpu-mfma-vgpr-form"]),
    # sparse pyramidal Lucas-Kanade flow of the flow tracker: float32 op by op as the scalar CPU code it restates
    ("flow.hip", ["-ffp-contract=off"]),
    ("tracker.hip", []),
    ("h264dec.hip", []),  # host code: the slice decoder behind MediaVideo
    ("network.hip", []),
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


# storage-type variants of the network kernels (csrc/bf16.h): name -> (library file, extra defines)
VARIANTS = {"bf16": (LIB, []), "fp16": (os.path.join(LIBDIR, "libsleap_amd_fp16.so"), ["-DSA_HALF_FP16=1"])}


def lib_path(dtype="bf16"):
    return VARIANTS[dtype][0]


def _build_variant(dtype, force, verbose):
    lib, defines = VARIANTS[dtype]
    objdir = LIBDIR if dtype == "bf16" else os.path.join(LIBDIR, dtype)
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    objs, cmds = [], []
    dep_mtime = max(os.path.getmtime(h) for h in _deps())
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), dep_mtime):
            cmds.append([hipcc, "-c", s, "-o", o] + COMMON + extra + defines)
    rebuilt = bool(cmds)
    if cmds:  # the translation units are independent: compile them concurrently (80 s -> ~25 s per variant)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)

        with ThreadPoolExecutor(max_workers=min(len(cmds), os.cpu_count() or 1, 8)) as ex:
            list(ex.map(run, cmds))
    if rebuilt or not os.path.exists(lib):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


def build(force=False, verbose=True, dtypes=("bf16", "fp16")):
    """Compile every storage-type variant (fp16 = the default the Python layer loads, bf16); returns the path of the bf16
    library file (`libsleap_amd.so`, the historical name)."""
    os.makedirs(LIBDIR, exist_ok=True)
    for d in dtypes:
        _build_variant(d, force, verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
