# kernel statistics of the BASELINE configurations that are not the bench line (configs[0], [1], [2], [4]) + flow tracker rate
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03o}; mkdir -p $O; cd $R
timeout 300 python tools/flow_bench.py 64 > $O/flow_bench.md 2> $O/flow_bench.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python tools/bench_configs.py 5 > $O/configs.md 2> $O/kt.log
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/other_configs_kernel_stats.md
rm -rf $O/kt/*/*.db.tmp; cat $O/flow_bench.md; tail -3 $O/flow_bench.err; cat $O/configs.md; head -30 $O/other_configs_kernel_stats.md | cut -c1-170
