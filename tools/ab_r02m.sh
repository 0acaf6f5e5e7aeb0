# flow tracker after the pyramid / Lucas-Kanade index-arithmetic rework: parity tests, standalone rate, e2e rate, kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02m; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_flow.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/flow_bench.py 64 > $O/flow_bench.md 2> $O/flow_bench.err; cat $O/flow_bench.md; tail -2 $O/flow_bench.err
timeout 300 python tools/predict_e2e.py 2560 arrays flow 2>&1 | tail -4
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python tools/flow_bench.py 64 > $O/flow_bench_prof.md 2> $O/kt.log
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/flow_kernel_stats.md
rm -rf $O/kt; head -16 $O/flow_kernel_stats.md | cut -c1-190
