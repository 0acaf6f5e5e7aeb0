"""Static instruction mix of the gfx950 code hipcc generates for one translation unit -- no GPU needed.

    python tools/isa_stats.py sleap_amd/csrc/stem16.hip [--fp16] [--filter gray] [extra hipcc flags ...]

Per kernel: VGPR / AGPR / LDS, and counts of VALU, MFMA, v_accvgpr_* (accumulators the compiler kept in AGPRs cost one of
these per value in an epilogue), LDS reads / writes, global / buffer memory instructions, s_barrier, s_waitcnt. The flags a
file is built with come from sleap_amd/build.py. This is how the VALU-bound conv0 loop of stem16 and its AGPR accumulators
were found (34 VALU instructions per 3 MFMAs).
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, ".")
from sleap_amd import build as B


def main(argv):
    src = argv[0]
    fp16 = "--fp16" in argv
    flt = argv[argv.index("--filter") + 1] if "--filter" in argv else ""
    extra = [a for a in argv[1:] if a not in ("--fp16", "--filter", flt)]
    flags = dict(B.SOURCES).get(os.path.basename(src), [])
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [B._hipcc(), "-S", "--cuda-device-only", src, "-o", out, "-O3", "-std=c++17", f"--offload-arch={B.ARCH}",
               "-I", os.path.join(os.path.dirname(B.CSRC), "..", "include"), "-I", B.CSRC] + flags + (["-DSA_HALF_FP16=1"] if fp16 else []) + extra
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    kernels = re.findall(r"^(_Z\w+):.*?\n(.*?)s_endpgm", text, flags=re.S | re.M)
    meta = {m[0]: m[1:] for m in re.findall(r"\.set (_Z\w+)\.num_vgpr, (\d+)\n\s*\.set \1\.num_agpr, (\d+)", text)}
    lds = dict(re.findall(r"\.amdhsa_kernel (_Z\w+)\n(?:.*\n)*?\s*\.amdhsa_group_segment_fixed_size (\d+)", text))
    print("| kernel | vgpr | agpr | static lds B | VALU | MFMA | accvgpr | ds_read | ds_write | vmem | barriers | waitcnt |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, body in kernels:
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
        if flt and flt not in demangled:
            continue
        ins = re.findall(r"^\s+([a-z_0-9]+)", body, flags=re.M)
        c = lambda pat: sum(1 for i in ins if re.match(pat, i))  # noqa: E731
        v, a = meta.get(name, ("?", "?"))
        print(f"| `{demangled[:90]}` | {v} | {a} | {lds.get(name, '?')} | {c(r'v_(?!mfma|accvgpr)')} | {c(r'v_mfma')} | {c(r'v_accvgpr')} | "
              f"{c(r'ds_read|ds_load')} | {c(r'ds_write|ds_store')} | {c(r'global_|buffer_|flat_')} | {c(r's_barrier')} | {c(r's_waitcnt')} |")


if __name__ == "__main__":
    main(sys.argv[1:])
