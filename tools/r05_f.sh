cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05f}; mkdir -p $O; cd $R
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o run -- python bench.py --no-cpu-baseline --no-extras --steps 30 > $O/kt.log 2>&1
k=$(find $O/kt -name "*kernel_trace.csv" | head -1); python tools/launch_gaps.py $k > $O/launch_gaps.txt; cat $O/launch_gaps.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt8 -o run -- python bench.py --no-cpu-baseline --no-extras --batch 8 --steps 100 > $O/kt8.log 2>&1
k=$(find $O/kt8 -name "*kernel_trace.csv" | head -1); python tools/launch_gaps.py $k > $O/launch_gaps_b8.txt; cat $O/launch_gaps_b8.txt
rm -rf $O/kt $O/kt8
