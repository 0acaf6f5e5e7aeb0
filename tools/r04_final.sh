# round 4: the checks the driver runs at round end -- the whole -m gpu suite, smoke(), the default bench line; with a second
# argument the profile set of the same code (kernel trace, MFMA / FETCH / WRITE counter passes, SQ counters, per-layer table,
# 8-frame kernel trace, the ResNet-50 per-launch table + its kernel trace and FETCH / WRITE passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04z}; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 6 $O/pytest.log | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.log; cut -c1-220 $O/bench_line.json; python -c "
import json; j=json.loads(open('$O/bench_line.json').readline()); print({k: j['roofline'][k] for k in ('frac','frac_forward','frac_materialised','frac_forward_materialised','frac_dense','traffic','network_ms_per_step')}); print(j['sustained']); print(j['literal_split_8_per_gpu']); print(j['cpu_baseline']['parity_vs_oracle'])"
if [ -n "$2" ]; then
timeout 300 python bench.py --no-cpu-baseline --no-extras --layers --steps 20 > $O/bench_layers.json 2> $O/layers.log
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python bench.py --no-cpu-baseline --no-extras > $O/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o run -- python bench.py --no-cpu-baseline --no-extras > $O/mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- python bench.py --no-cpu-baseline --no-extras > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o run -- python bench.py --no-cpu-baseline --no-extras > $O/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_b8 -o run -- python bench.py --no-cpu-baseline --no-extras --batch 8 --steps 40 > $O/kt_b8.log 2>&1
for d in kt kt_b8; do db=$(find $O/$d -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/${d}_kernel_stats.md; done
f=$(find $O/mfma -name "*counter_collection.csv" | head -1); k=$(find $O/mfma -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/pmc_mfma.py $f $k > $O/pmc_mfma_util.md 2>&1
ff=$(find $O/fetch -name "*counter_collection.csv" | head -1); fw=$(find $O/write -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_traffic.py $ff $fw 0 $O/pmc_hbm_traffic.json > $O/pmc_hbm_traffic.md 2>&1
rm -rf $O/kt $O/kt_b8 $O/mfma $O/fetch $O/write
bash tools/prof_sq.sh ${1:-r04z}/sq > $O/prof_sq.log 2>&1; cp $O/sq/sq.md $O/pmc_sq_counters.md; rm -rf $O/sq/a $O/sq/b
# ResNet-50 (configs[4] architecture): per-launch table, kernel trace, FETCH / WRITE passes
timeout 300 python tools/net_profile.py resnet50 1024 16 0.012 > $O/resnet50_per_launch.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/rkt -o run -- python tools/net_profile.py resnet50 1024 16 1.0 > $O/rkt.log 2>&1
db=$(find $O/rkt -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/resnet50_kernel_stats.md
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/rfetch -o run -- python tools/net_profile.py resnet50 1024 16 1.0 > $O/rfetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/rwrite -o run -- python tools/net_profile.py resnet50 1024 16 1.0 > $O/rwrite.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/rmfma -o run -- python tools/net_profile.py resnet50 1024 16 1.0 > $O/rmfma.log 2>&1
ff=$(find $O/rfetch -name "*counter_collection.csv" | head -1); fw=$(find $O/rwrite -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_traffic_generic.py $ff $fw > $O/resnet50_pmc_hbm_traffic.md 2>&1
f=$(find $O/rmfma -name "*counter_collection.csv" | head -1); k=$(find $O/rmfma -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/pmc_mfma.py $f $k > $O/resnet50_pmc_mfma_util.md 2>&1
rm -rf $O/rkt $O/rfetch $O/rwrite $O/rmfma
timeout 400 python tools/bench_configs.py 10 > $O/other_configs.md 2> $O/other_configs.err; cat $O/other_configs.md | cut -c1-200
ls $O; head -30 $O/kt_kernel_stats.md | cut -c1-200
fi
