"""Pivot a rocprofv3 counter_collection.csv: per kernel (name prefix filter) mean counter values."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "conv3x3"
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
with open(path) as f:
    for r in csv.DictReader(f):
        if flt not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0][-60:]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[(name, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for name, cs in acc.items():
    d = [v for (n, _), v in dur.items() if n == name]
    print(f"== {name}: {len(d)} dispatches, mean {sum(d) / len(d) / 1e3:.1f} us")
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v) / len(v):16.1f}")
