cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02g; O=gpurun_out/r02g; L=$GRAFT_REPO_ROOT/sleap_amd/lib
timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_persistent.py tests/test_gpu_fp16.py tests/test_gpu_fullsize.py tests/test_gpu_benchmark_parity.py tests/test_gpu_backbones.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
for v in prev new; do
  unset SLEAP_AMD_LIB_FP16
  case $v in new) ;; prev) export SLEAP_AMD_LIB_FP16=$L/alt_prev_epilogue.so;; esac
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['network_ms_per_step'])"
done; done
paste <(grep "ms " $O/layers_prev_2.log | cut -c1-60) <(grep "ms " $O/layers_new_2.log | cut -c45-60)
