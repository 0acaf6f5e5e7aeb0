# round 3: HBM traffic counters of the configs[4] network's launches (two separate --pmc passes, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03pr}; mkdir -p $O; cd $R
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- python tools/net_profile.py resnet50 1024 16 > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o run -- python tools/net_profile.py resnet50 1024 16 > $O/write.log 2>&1
python - <<PY
import csv, collections, glob
def load(d, name):
    f = glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
            a[(k, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return a
f, w = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
print("| kernel | grid | launches | FETCH_SIZE x 2 [MB / launch] | WRITE_SIZE [MB / launch] |")
print("|---|---|---|---|---|")
for k in sorted(f, key=lambda k: -sum(f[k])):
    if sum(f[k]) / 1024 < 50: continue
    print(f"| {k[0][:60]} | {k[1]} | {len(f[k])} | {sum(f[k]) / len(f[k]) / 1024 * 2:.1f} | {sum(w.get(k, [0])) / max(len(w.get(k, [1])), 1) / 1024:.1f} |")
PY
rm -rf $O/fetch $O/write
