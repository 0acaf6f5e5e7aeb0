"""For the large host-to-device copies of a rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace run (csv): when the
copy was ISSUED (its hipMemcpy* call) against when it RAN, and what the GPU did in between.  python tools/copy_issue_vs_run.py <dir>"""
import csv
import glob
import sys

d = sys.argv[1]
f = lambda pat: glob.glob(d + "/**/*" + pat, recursive=True)  # noqa: E731
K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f("kernel_trace.csv")[0])))
M = [r for r in csv.DictReader(open(f("memory_copy_trace.csv")[0]))]
api = {}
for r in csv.DictReader(open(f("hip_api_trace.csv")[0])):
    if "Memcpy" in r["Function"]:
        api[r["Correlation_Id"]] = (r["Function"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
big = [r for r in M if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 300e3][-8:]
for r in big:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    fn, a0, a1 = api.get(r["Correlation_Id"], ("?", s, s))
    busy = sum(max(0, min(ke, s) - max(ks, a0)) for ks, ke in K)
    print(f"{fn}: call took {(a1 - a0) / 1e3:.0f} us; copy started {(s - a0) / 1e6:.3f} ms after the call, ran {(e - s) / 1e6:.3f} ms; "
          f"kernels were running for {busy / 1e6:.3f} ms of that wait")
