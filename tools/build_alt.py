"""Build an alternative fp16 library for A/B runs: one or more translation units recompiled with extra flags, linked with the
standard objects of the other files.   python tools/build_alt.py <name.so> <file.hip>[,<file2.hip>] <flags...>
The result lands in sleap_amd/lib/<name.so> (travels to the GPU box; load it with SLEAP_AMD_LIB_FP16=<path>)."""
import os
import subprocess
import sys

sys.path.insert(0, ".")
from sleap_amd import build as B

name, files, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
B.build(verbose=False)
objdir = os.path.join(B.LIBDIR, "fp16")
objs = []
for src, extra in B.SOURCES:
    o = os.path.join(objdir, src.replace(".hip", ".o"))
    if src in files:
        o = os.path.join(objdir, name + "." + src.replace(".hip", ".o"))
        subprocess.check_call([B._hipcc(), "-c", os.path.join(B.CSRC, src), "-o", o] + B.COMMON + extra + ["-DSA_HALF_FP16=1"] + flags)
    objs.append(o)
out = os.path.join(B.LIBDIR, name)
subprocess.check_call([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", out] + objs)
print(out)
