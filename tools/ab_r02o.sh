# flow tracker: 4-pixel pyramid kernels, stage margin 4, shared tracker stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02p; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_flow.py tests/test_gpu_inference.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/flow_bench.py 64 > $O/flow_bench.md 2> $O/flow_bench.err; cat $O/flow_bench.md; tail -2 $O/flow_bench.err
for trk in flow none flow flowmaxtracks; do
  if [ $trk = none ]; then a=""; else a=$trk; fi
  timeout 300 python tools/predict_e2e.py 2560 arrays $a 2>&1 | grep "frames/s" | tail -2
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python tools/flow_bench.py 64 > $O/flow_bench_prof.md 2> $O/kt.log
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/flow_kernel_stats.md
rm -rf $O/kt; head -16 $O/flow_kernel_stats.md | cut -c1-190
