"""GPU timeline of a rocprofv3 --kernel-trace --memory-copy-trace run (csv output): busy time of kernels and copies over the last
`window_ms` of the run, the overlap between them, and the largest idle gaps.   python tools/timeline_gaps.py <dir> [window_ms]"""
import csv
import glob
import sys

d = sys.argv[1]
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 140e6
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
mc = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in csv.DictReader(open(kt))]
M = []
if mc:
    for r in csv.DictReader(open(mc[0])):
        M.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?")))
end = max(e for _, e, _ in K)
K = sorted(k for k in K if k[0] >= end - win)
M = sorted(m for m in M if m[0] >= end - win)


def union(iv):
    out, cur = [], None
    for s, e in sorted(iv):
        if cur and s <= cur[1]:
            cur[1] = max(cur[1], e)
        else:
            cur = [s, e]
            out.append(cur)
    return out


ku = union([(s, e) for s, e, _ in K])
busy = sum(e - s for s, e in ku)
print(f"window {win / 1e6:.0f} ms: {len(K)} kernels busy {busy / 1e6:.1f} ms (sum of durations {sum(e - s for s, e, _ in K) / 1e6:.1f}); {len(M)} copies")
by = {}
for s, e, dr in M:
    by.setdefault(dr, []).append(e - s)
for dr, v in by.items():
    print(f"  copies {dr}: n {len(v)} total {sum(v) / 1e6:.2f} ms, max {max(v) / 1e6:.3f} ms")
big = [(s, e, dr) for s, e, dr in M if e - s > 200e3]
ov = 0
for s, e, dr in big:
    for a, b in ku:
        ov += max(0, min(e, b) - max(s, a))
print(f"  large copies: {len(big)}, {sum(e - s for s, e, _ in big) / 1e6:.2f} ms, of which under kernels {ov / 1e6:.2f} ms")
gaps = sorted(((ku[i + 1][0] - ku[i][1], ku[i][1]) for i in range(len(ku) - 1)), reverse=True)[:12]
print("  largest idle gaps (ms):", [round(g / 1e6, 3) for g, _ in gaps])
print(f"  idle inside the window: {(ku[-1][1] - ku[0][0] - busy) / 1e6:.1f} ms of {(ku[-1][1] - ku[0][0]) / 1e6:.1f}")
