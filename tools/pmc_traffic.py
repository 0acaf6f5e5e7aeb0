"""HBM traffic of the conv kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py.
FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md §HBM).

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <n B=64 forwards, 0 = count them> [out.json]

With `out.json` the per-step total is also written as the small JSON file bench.py reads `roofline.traffic` from
(profiles/r<NN>_pmc_hbm_traffic.json: {"conv_family_bytes_per_step", "frames_per_step", "size", "source"}).
"""
import csv
import re
import sys
from collections import defaultdict

PAT = re.compile(r"(conv3x3_dma_kernel<[^>]*>|stem16_kernel<\d>|stem16_gray_kernel|convpair_16_32_32_kernel|convpair_persist_kernel|upsample2x_kernel|upsample2x_bilinear_block_kernel|upsample2x_bilinear_c16_kernel<\d>)")


def load(path, counter):
    a = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = PAT.search(r["Kernel_Name"])
        if m:
            a[(m.group(1), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return a


def passes(a):
    """Forward passes in one profiled run = launches of the once-per-pass kernels (the smallest launch count of the family):
    bench.py warms up by TIME, so the count differs from run to run and between the FETCH and the WRITE pass."""
    return min(len(v) for v in a.values())


def main(fetch_csv, write_csv, n_fwd=0, out_json=None):
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    nf, nw = (n_fwd, n_fwd) if n_fwd else (passes(f), passes(w))
    print("| kernel | grid | launches (FETCH pass) | FETCH_SIZE x2 [MiB/launch] | WRITE_SIZE [MiB/launch] |")
    print("|---|---|---|---|---|")
    tf = tw = 0.0
    for key in sorted(f, key=lambda k: -sum(f[k])):
        fv = sum(f[key]) / len(f[key]) / 1024 * 2
        wv = sum(w.get(key, [0])) / max(len(w.get(key, [1])), 1) / 1024
        print(f"| `{key[0]}` | {key[1]} | {len(f[key])} | {fv:.1f} | {wv:.1f} |")
        if "upsample" not in key[0]:
            tf += sum(f[key]) / 1024 * 2 / nf
            tw += sum(w.get(key, [0])) / 1024 / nw
    print(f"\nconv family (conv3x3_dma_kernel + convpair + stem16): {nf} forward passes of 64 frames in the FETCH run, {nw} in the WRITE run "
          "(bench.py --no-cpu-baseline --no-extras: time-based pre-warm + 3 warm-up + 10 timed steps + instrumented passes)")
    gb = 1048576 / 1e9  # the counters are KiB; the table is MiB; these sums are decimal GB / MB, as the bench line's `traffic`
    print(f"-> per step (all conv-family launches of one forward pass, 64 frames): FETCH x2 {tf * gb:.2f} GB + WRITE {tw * gb:.2f} GB = "
          f"{(tf + tw) * gb:.2f} GB = {(tf + tw) * gb * 1000 / 64:.0f} MB/frame (decimal; = {(tf + tw) / 1024:.2f} GiB); algorithmic bytes of the "
          "current plan: DeviceNetwork.op_bytes (DESIGN.md section 5)")

    if out_json:
        import json

        json.dump({"conv_family_bytes_per_step": (tf + tw) * 1048576, "fetch_x2_bytes_per_step": tf * 1048576,
                   "write_bytes_per_step": tw * 1048576, "frames_per_step": 64, "size": 1024, "forward_passes": [nf, nw],
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH doubled (gfx950), WRITE uncalibrated"},
                  open(out_json, "w"))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0, sys.argv[4] if len(sys.argv) > 4 else None)
