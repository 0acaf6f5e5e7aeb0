"""Idle time BETWEEN the network's launches of a step, from a rocprofv3 --kernel-trace csv of bench.py: for consecutive network kernels
(stem16, convpair, conv3x3_dma, upsample2x) the gap end(i) -> start(i + 1), per position in the step.
    python tools/launch_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
net = sorted(r for r in rows if any(k in r[2] for k in ("stem16", "convpair", "conv3x3_dma", "upsample2x_bilinear_c16")))
steps, cur = [], []
for r in net:
    if "stem16" in r[2] and cur:
        steps.append(cur)
        cur = []
    cur.append(r)
steps.append(cur)
steps = [s for s in steps if len(s) == 18][-40:]  # full forward passes of the benchmark plan, the last 40
gap, dur = defaultdict(list), defaultdict(list)
for s in steps:
    for i, (a, b, n) in enumerate(s):
        dur[i].append(b - a)
        if i + 1 < len(s):
            gap[i].append(s[i + 1][0] - b)
print(f"{len(steps)} forward passes of 18 launches")
tg = td = 0.0
for i in range(18):
    d = sum(dur[i]) / len(dur[i]) / 1e3
    g = sum(gap[i]) / len(gap[i]) / 1e3 if gap[i] else 0.0
    td += d
    tg += g
    print(f"launch {i:2d}: duration {d:8.1f} us, gap to the next launch {g:6.2f} us")
between = [steps[k + 1][0][0] - steps[k][-1][1] for k in range(len(steps) - 1)]
print(f"sum of durations {td / 1e3:.3f} ms, sum of the 17 gaps inside a pass {tg / 1e3:.3f} ms, last launch -> next pass's first launch {sum(between) / max(len(between), 1) / 1e3:.2f} us")
