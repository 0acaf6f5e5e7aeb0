cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02k; O=gpurun_out/r02k
timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_persistent.py tests/test_gpu_backbones.py tests/test_gpu_topdown.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
for v in off on; do
  if [ $v = off ]; then export SA_CONV_SMALL_MT1=0; else unset SA_CONV_SMALL_MT1; fi
  for b in 8 16 64; do
  timeout 200 python bench.py --batch $b --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', 'B=$b', j['value'], j['ms_per_step'], j['roofline']['frac'])"
  done
done; done
