"""Matrix-core utilisation per kernel from one rocprofv3 --pmc pass over bench.py:

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> -- python bench.py ...
    python tools/pmc_mfma.py <counter_collection.csv> [kernel_trace.csv]

SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over all SIMDs (= 32 x N_mfma for v_mfma_f32_32x32x16_bf16,
MI355X_MICROARCH.md "Per-instruction cycle constants"); GRBM_GUI_ACTIVE counts the busy cycles of the launch summed
over the 8 XCDs (checked against the kernel trace: GUI_ACTIVE / duration = 17.1 cycles/ns = 8 x 2.14 GHz on the conv
kernels, 8 x 2.5 GHz on light kernels). With 256 CUs x 4 SIMDs:

    utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024)     -- independent of the clock the chip settled at
    effective clock = GUI_ACTIVE / 8 / duration           -- needs the kernel trace of the same run

The derived clock is only printed for launches of >= 100 us (round 5): GUI_ACTIVE covers the command processor's work around a
dispatch too, and on short launches the quotient came out at 3.1-3.6 GHz -- impossible on this part.
"""
import csv
import re
import sys
from collections import defaultdict

PAT = re.compile(r"(conv3x3_dma_kernel<[^>]*>|conv3x3_mfma_kernel<[^>]*>|stem16_kernel<\d>|stem16_gray_kernel|convpair_16_32_32_kernel|convpair_persist_kernel|tapconv_kernel<[^>]*>|tapconv_coloop_kernel<[^>]*>|imgconv_mfma_kernel<[^>]*>|conv1x1_head_mfma_kernel)")


def main(path, trace=None):
    busy, act, clk, durs = defaultdict(list), defaultdict(list), defaultdict(list), defaultdict(list)
    dur = {}
    if trace:
        for r in csv.DictReader(open(trace)):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for r in csv.DictReader(open(path)):
        m = PAT.search(r["Kernel_Name"])
        if not m:
            continue
        key = (m.group(1), int(r["Grid_Size"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
            clk[key].append(float(r["Counter_Value"]) / 8 / dur[r["Dispatch_Id"]])
            durs[key].append(dur[r["Dispatch_Id"]])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            busy[key].append(float(r["Counter_Value"]))
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            act[key].append(float(r["Counter_Value"]))
    print("| kernel | grid | launches | MFMA busy cycles / launch | GUI_ACTIVE / 8 cycles / launch | MFMA utilisation | clock GHz |")
    print("|---|---|---|---|---|---|---|")
    tb = ta = 0.0
    for key in sorted(busy, key=lambda k: -sum(busy[k])):
        b, a = sum(busy[key]) / len(busy[key]), sum(act[key]) / max(len(act[key]), 1) / 8
        ck = f"{sum(clk[key]) / len(clk[key]):.2f}" if clk[key] else "-"
        if durs[key] and sum(durs[key]) / len(durs[key]) < 100e3:  # ns
            ck = "n/a (< 100 us)"
        print(f"| `{key[0]}` | {key[1]} | {len(busy[key])} | {b:.4g} | {a:.4g} | {b / (a * 1024):.3f} | {ck} |")
        tb += sum(busy[key])
        ta += sum(act[key]) / 8
    print(f"\nall MFMA conv launches of the run: MFMA busy {tb:.4g} cycles over {ta:.4g} GPU-active cycles x 1024 SIMDs "
          f"-> utilisation {tb / (ta * 1024):.3f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
