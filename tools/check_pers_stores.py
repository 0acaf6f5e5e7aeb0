"""Build-time check of the store-count model behind the PERS kernels' counted wait (csrc/conv3x3.hip, `stores_behind`).

The persistent two-workgroup kernels (off by default, A/B only) wait for the next tile's first copies with `s_waitcnt vmcnt(S)`,
S = the store instructions the epilogue issued after them. The model: ONE global_store per valid 16-channel piece and output row
(MT x 2 pieces x (R full-resolution + R / 2 pooled rows)). If a compiler change merged or split stores, a tile could be computed
from an LDS stage whose copy is still in flight -- silently. This script compiles the translation unit to ISA (no GPU needed)
and compares the STATIC count of global_store instructions in every PERS instantiation with the model.

    python tools/check_pers_stores.py        -> prints one line per kernel, exit code 1 on a mismatch
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sleap_amd import build as B


def pers_store_counts(fp16=True):
    src = os.path.join(B.CSRC, "conv3x3.hip")
    flags = dict(B.SOURCES)["conv3x3.hip"]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([B._hipcc(), "-S", "--cuda-device-only", src, "-o", out, "-O3", "-std=c++17", f"--offload-arch={B.ARCH}",
                        "-I", os.path.join(os.path.dirname(B.CSRC), "..", "include"), "-I", B.CSRC] + flags +
                       (["-DSA_HALF_FP16=1"] if fp16 else []), check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    res = []
    for name, body in re.findall(r"^(_Z\w+):.*?\n(.*?)s_endpgm", text, flags=re.S | re.M):
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        m = re.search(r"conv3x3_dma_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\w+), (\d+), (\w+), (\w+), (-?\d+), (\w+), (\w+)>", dem)
        if not m or m.group(12) != "true":
            continue
        mt, r = int(m.group(1)), int(m.group(4))
        n = len(re.findall(r"^\s+global_store_", body, flags=re.M))
        res.append((m.group(0), n, mt * 2 * (r + r // 2)))
    return res


if __name__ == "__main__":
    bad = 0
    for name, got, want in pers_store_counts():
        print(f"{name}: {got} global_store instructions, model {want}" + ("" if got == want else "   <-- MISMATCH"))
        bad += got != want
    sys.exit(1 if bad else 0)
