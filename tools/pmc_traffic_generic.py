"""HBM traffic per launch of every kernel of a profiled run, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel trace
only). FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE is
uncalibrated there. Counter units are KiB; the table is decimal MB.

    python tools/pmc_traffic_generic.py <fetch counter_collection.csv> <write counter_collection.csv> [min MB per launch, default 20]
"""
import collections
import csv
import sys


def load(path, name):
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
            a[(k, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return a


def main(fetch_csv, write_csv, min_mb=20.0):
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    mb = 1024 / 1e6
    print("| kernel | grid | launches | FETCH_SIZE x 2 [MB / launch] | WRITE_SIZE [MB / launch] |")
    print("|---|---|---|---|---|")
    tf = tw = 0.0
    for k in sorted(set(f) | set(w), key=lambda k: -(sum(f.get(k, [0])) * 2 + sum(w.get(k, [0])))):
        fv = sum(f.get(k, [0])) / max(len(f.get(k, [1])), 1) * 2 * mb
        wv = sum(w.get(k, [0])) / max(len(w.get(k, [1])), 1) * mb
        tf += sum(f.get(k, [0])) * 2 * mb
        tw += sum(w.get(k, [0])) * mb
        if fv + wv >= min_mb:
            print(f"| `{k[0][:80]}` | {k[1]} | {len(f.get(k, w.get(k, [])))} | {fv:.1f} | {wv:.1f} |")
    print(f"\nall launches of the run: FETCH x 2 {tf / 1e3:.2f} GB + WRITE {tw / 1e3:.2f} GB")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 20.0)
