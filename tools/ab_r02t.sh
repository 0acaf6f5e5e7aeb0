# mid-chunk issue of the next chunk's copies: default rule (tap 4 for layers with >= 5 chunks) vs off vs other taps / spread issue
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02t; O=gpurun_out/r02t
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_persistent.py tests/test_gpu_fp16.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
for v in new off itap2 itap3 itap5 spread1 spread2 min8; do
  unset SLEAP_AMD_LIB_FP16 SA_CONV_LATE_ISSUE
  case $v in new) ;; off) export SA_CONV_LATE_ISSUE=0;; *) export SLEAP_AMD_LIB_FP16=$PWD/sleap_amd/lib/alt_$v.so;; esac
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['frac'])"
done; done
paste <(grep "ms " $O/layers_off_2.log | cut -c1-60) <(grep "ms " $O/layers_new_2.log | cut -c45-60) <(grep "ms " $O/layers_itap2_2.log | cut -c45-60) <(grep "ms " $O/layers_itap3_2.log | cut -c45-60) <(grep "ms " $O/layers_itap5_2.log | cut -c45-60) <(grep "ms " $O/layers_spread1_2.log | cut -c45-60) <(grep "ms " $O/layers_spread2_2.log | cut -c45-60) <(grep "ms " $O/layers_min8_2.log | cut -c45-60)
