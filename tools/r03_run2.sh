# round 3, GPU call 2 (first of the re-entered session): the whole -m gpu suite (no -x: every failure in one call), the bench
# line, the per-layer table and a kernel trace of the current tree.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03b}; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 40 $O/pytest.log
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.log; cut -c1-1800 $O/bench_line.json; tail -n 3 $O/bench_err.log
timeout 300 python bench.py --no-cpu-baseline --no-extras --layers --steps 20 > $O/bench_layers.json 2> $O/layers.log; tail -n 30 $O/layers.log
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python bench.py --no-cpu-baseline --no-extras > $O/kt.log 2>&1
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/kt_kernel_stats.md; rm -rf $O/kt
head -n 40 $O/kt_kernel_stats.md
