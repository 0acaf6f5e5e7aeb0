# flow tracker on a stream of its own + no synchronisation between the pyramids and the Lucas-Kanade launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02n; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_flow.py tests/test_gpu_inference.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/flow_bench.py 64 > $O/flow_bench.md 2> $O/flow_bench.err; head -5 $O/flow_bench.md; tail -2 $O/flow_bench.err
for trk in flow none flow simple flowmaxtracks; do
  if [ $trk = none ]; then a=""; else a=$trk; fi
  timeout 300 python tools/predict_e2e.py 2560 arrays $a 2>&1 | grep "frames/s" | tail -2
done
