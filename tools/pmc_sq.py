"""Per-kernel table of arbitrary rocprofv3 --pmc counters (mean per launch), keyed by kernel template + grid size.

    python tools/pmc_sq.py <counter_collection.csv> [more.csv ...] [--min-grid N]
"""
import csv
import re
import sys
from collections import defaultdict

PAT = re.compile(r"([A-Za-z_0-9]+_kernel(?:<[^>]*>)?)")


def main(paths, min_grid=100000):
    vals = defaultdict(lambda: defaultdict(list))
    names = []
    for path in paths:
        for r in csv.DictReader(open(path)):
            m = PAT.search(r["Kernel_Name"])
            if not m or int(r["Grid_Size"]) < min_grid:
                continue
            key = (m.group(1), int(r["Grid_Size"]))
            vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] not in names:
                names.append(r["Counter_Name"])
    print("| kernel | grid | n | " + " | ".join(names) + " |")
    print("|---|---|---|" + "---|" * len(names))
    for key in sorted(vals, key=lambda k: -sum(vals[k].get(names[0], [0]))):
        row = [f"{sum(vals[key][n]) / len(vals[key][n]):.4g}" if vals[key].get(n) else "-" for n in names]
        n = len(vals[key][names[0]]) if vals[key].get(names[0]) else 0
        print(f"| `{key[0]}` | {key[1]} | {n} | " + " | ".join(row) + " |")


if __name__ == "__main__":
    argv = sys.argv[1:]
    mg = 100000  # (persistent kernels launch one workgroup per CU: 131072-262144 threads)
    if "--min-grid" in argv:
        i = argv.index("--min-grid")
        mg = int(argv[i + 1])
        del argv[i:i + 2]
    main([x for x in argv if not x.startswith("--")], mg)
