# A/B of library builds by kernel trace of the whole bench step (every kernel's average over ~100 launches), alternating on one box.
#   bash tools/kt_ab.sh <out-name> <alt1.so> ...   (alternates made by tools/build_alt.py; "base" = the default build)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; L=$R/sleap_amd/lib; O=$R/gpurun_out/${1:-ktab}; shift; mkdir -p $O; cd $R
[ -n "$KT_TESTS" ] && timeout 900 python -m pytest $KT_TESTS -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 3 | cut -c1-200 | tee $O/tests.txt
for i in 1 2 3; do
for v in base "$@"; do
  if [ $v = base ]; then unset SLEAP_AMD_LIB_FP16; else export SLEAP_AMD_LIB_FP16=$L/$v; fi
  rm -rf $O/kt; timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python bench.py --no-cpu-baseline --no-extras --steps 40 ${KT_BENCH_ARGS:-} > $O/kt_${v}_$i.log 2>&1
  db=$(find $O/kt -name "*.db" | head -1); python tools/rocpd_stats.py $db > $O/stats_${v}_$i.md
  echo "$v $i $(grep -o '"value": [0-9.]*' $O/kt_${v}_$i.log | head -1) $(grep -o '"result_digest": "[0-9a-f]*"' $O/kt_${v}_$i.log) $(grep 'total GPU' $O/stats_${v}_$i.md)" | tee -a $O/ab.txt
done; done
rm -rf $O/kt
python - "$O" base "$@" <<'PY'
import sys, re, glob, collections
O, names = sys.argv[1], sys.argv[2:]
tab = collections.defaultdict(dict)
for v in names:
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/stats_{v}_*.md"):
        for l in open(f):
            c = [x.strip() for x in l.split("|")]
            if len(c) > 12 and c[2].isdigit():
                acc[(c[1][:70], c[11])].append(float(c[4]))
    for k, xs in acc.items():
        tab[k][v] = sum(xs) / len(xs)
print("| kernel (grid) | " + " | ".join(names) + " |   avg us over the runs")
for k in sorted(tab, key=lambda k: -max(tab[k].values())):
    if max(tab[k].values()) > 20:
        print(f"| {k[0]} ({k[1]}) | " + " | ".join(f"{tab[k].get(v, float('nan')):.1f}" for v in names) + " |")
PY
