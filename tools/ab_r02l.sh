cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02l; O=gpurun_out/r02l
SA_CONV_HEAD_SPLIT=1 timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_persistent.py tests/test_gpu_fp16.py tests/test_gpu_benchmark_parity.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do
for v in base split; do
  unset SA_CONV_HEAD_SPLIT
  case $v in split) export SA_CONV_HEAD_SPLIT=1;; esac
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['frac'])"
  grep "head24" $O/layers_${v}_$i.log | cut -c1-62
done; done
