cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02td; mkdir -p $O; cd $R
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python tools/predict_e2e_topdown.py 1024 64 > $O/td.log 2>&1
db=$(find $O/kt -name "*.db" | head -1); python tools/rocpd_stats.py $db > $O/topdown_kernel_stats.md; rm -rf $O/kt
grep "frames/s" $O/td.log | tail -2 | cut -c1-110; head -24 $O/topdown_kernel_stats.md | cut -c1-175
