# round-2 session c: full GPU tests on the plane layout, bench NHWC vs planes alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02c; O=gpurun_out/r02c
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_tests.log; cat $O/gpu_tests.log
for i in 1 2; do
for v in nhwc planes16; do
  SA_LAYOUT=$v timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['network_ms_per_step'])"
done; done
paste <(grep "ms " $O/layers_nhwc_2.log | cut -c1-60) <(grep "ms " $O/layers_planes16_2.log | cut -c45-60)
