"""Per-LAYER counter table of conv3x3_dma_kernel (what bounds the dominant kernel), non-intrusive: the product
library under rocprofv3 --pmc, one layer shape of the benchmark plan after the other, a fixed number of launches each.

    rocprofv3 --kernel-trace --pmc <counters...> --output-format csv -d <dir> -o run -- python tools/pmc_layers.py run [B]
    python tools/pmc_layers.py report <counter_collection.csv> [more passes ...] > profiles/r05_pmc_dominant_per_layer.md

`run` launches every layer LAUNCHES times (no other conv3x3_dma launches in the process), so `report` can cut the dispatch
sequence of that kernel family into layers by position. Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
count quad-cycles per wave, SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs.
"""
import csv
import sys
from collections import defaultdict

LAUNCHES = 12
# (C0, C1, Cout, H, pooled) of the plain conv3x3_dma launches of the benchmark plan, in plan order
LAYERS = [(32, 0, 64, 256, False), (64, 0, 64, 256, True), (64, 0, 128, 128, False), (128, 0, 128, 128, True), (128, 0, 256, 64, False),
          (256, 0, 256, 64, True), (256, 0, 512, 32, False), (512, 0, 512, 32, False), (256, 512, 256, 64, False), (256, 0, 256, 64, False),
          (128, 256, 128, 128, False)]


def name_of(layer):
    C0, C1, Cout, H, pooled = layer
    return f"{C0}{'+' + str(C1) if C1 else ''}->{Cout} @{H}" + (" +pool" if pooled else "")


def run(B):
    import torch

    sys.path.insert(0, ".")
    from sleap_amd import _lib, ops

    g = torch.Generator().manual_seed(0)
    for C0, C1, Cout, H, pooled in LAYERS:
        k = (torch.randn((3, 3, C0 + C1, Cout), generator=g) * (2.0 / (9 * (C0 + C1))) ** 0.5).numpy()
        pw = ops.pack_conv3x3_weights(k, C0, C1, dtype="fp16")
        coutp = ops.pad16(Cout)
        bias = torch.zeros((coutp,), device="cuda")
        # post-ReLU-like activations (half zeros): what a layer of the network reads
        x0 = torch.randn((B, H, H, C0), device="cuda").clamp_(min=0).to(torch.float16)
        x1 = torch.randn((B, H, H, C1), device="cuda").clamp_(min=0).to(torch.float16) if C1 else None
        mode = (1 if C1 else 0) | _lib.LAYOUT_PLANES16
        for _ in range(LAUNCHES):
            ops.conv3x3(x0, x1, mode, pw, bias, coutp, True, (H, H), full=True, pooled=pooled)
        torch.cuda.synchronize()
        del x0, x1


def report(paths, B=64):
    per = [defaultdict(list) for _ in LAYERS]
    for path in paths:
        rows = defaultdict(dict)  # dispatch id -> {counter: value}
        for r in csv.DictReader(open(path)):
            if "conv3x3_dma_kernel" not in r["Kernel_Name"]:
                continue
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        ids = sorted(rows)
        if len(ids) != LAUNCHES * len(LAYERS):
            print(f"<!-- {path}: {len(ids)} conv3x3_dma dispatches, expected {LAUNCHES * len(LAYERS)}: skipped -->")
            continue
        for li in range(len(LAYERS)):
            for d in ids[li * LAUNCHES + 2:(li + 1) * LAUNCHES]:  # (the first two launches of a shape warm the caches)
                for c, v in rows[d].items():
                    per[li][c].append(v)
    n_simd = 256 * 4
    print(f"# `conv3x3_dma_kernel` per layer of the benchmark plan: SQ counters of the product build ({B} frames, fp16 storage, planes)\n")
    print("mean per launch over the last 10 of 12 launches of each shape; `busy` = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 "
          "SIMDs); shares are of the waves' resident cycles (SQ_WAVE_CYCLES): `parked` = SQ_WAIT_ANY (s_waitcnt / barrier), `issue stall` = "
          "SQ_WAIT_INST_ANY (of which `lds` = SQ_WAIT_INST_LDS), `issuing` = SQ_ACTIVE_INST_ANY; `waves / SIMD` = resident waves "
          "averaged over the launch (4 = two workgroups per CU all the time).\n")
    have = set()
    for p_ in per:
        have |= set(p_)
    cols = ["layer", "GFLOP", "kcycles / XCD", "TFLOP/s @2.4 GHz-equivalent", "MFMA busy", "waves / SIMD", "parked", "issue stall", "(lds)", "issuing"]
    extra = [c for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_LDS_IDX_ACTIVE",
                         "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU") if c in have]
    print("| " + " | ".join(cols + [e.replace("SQ_", "").lower() + " / wave-cycle" for e in extra]) + " |")
    print("|" + "---|" * (len(cols) + len(extra)))

    def mean(li, c):
        v = per[li].get(c)
        return sum(v) / len(v) if v else None

    for li, layer in enumerate(LAYERS):
        C0, C1, Cout, H, pooled = layer
        fl = 2.0 * B * H * H * (C0 + C1) * Cout * 9
        gui, busy, wc = mean(li, "GRBM_GUI_ACTIVE"), mean(li, "SQ_VALU_MFMA_BUSY_CYCLES"), mean(li, "SQ_WAVE_CYCLES")
        if not gui or not wc:
            continue
        cyc = gui / 8.0  # cycles of the launch (per XCD)

        def share(c):
            v = mean(li, c)
            return f"{v / wc:.3f}" if v is not None else "-"

        row = [name_of(layer), f"{fl / 1e9:.0f}", f"{cyc / 1e3:.0f}", f"{fl / (cyc / 2.4e9) / 1e12:.0f}",
               f"{busy / (cyc * n_simd):.3f}" if busy else "-", f"{wc * 4 / (cyc * n_simd):.2f}", share("SQ_WAIT_ANY"), share("SQ_WAIT_INST_ANY"),
               share("SQ_WAIT_INST_LDS"), share("SQ_ACTIVE_INST_ANY")]
        row += [share(e) for e in extra]
        print("| " + " | ".join(row) + " |")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 64)
    else:
        report(sys.argv[2:])
