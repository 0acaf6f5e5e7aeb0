cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02j; O=gpurun_out/r02j; L=$GRAFT_REPO_ROOT/sleap_amd/lib
timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_fp16.py tests/test_gpu_network_pin.py -m gpu -x -q 2>&1 | tail -2
SA_STEM16_PERSIST=5 timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_fp16.py tests/test_gpu_network_pin.py tests/test_gpu_benchmark_parity.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
for v in prev p0 p4 p5 p10; do
  unset SLEAP_AMD_LIB_FP16 SA_STEM16_PERSIST
  case $v in prev) export SLEAP_AMD_LIB_FP16=$L/alt_prev_stem.so;; p0) ;; p4) export SA_STEM16_PERSIST=4;; p5) export SA_STEM16_PERSIST=5;; p10) export SA_STEM16_PERSIST=10;; esac
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['frac'])"
  grep "stem" $O/layers_${v}_$i.log | cut -c1-62
done; done
