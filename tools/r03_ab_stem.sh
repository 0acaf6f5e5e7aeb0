# round 3: stem16 LDS layout (8-channel planes) and folded conv0 terms, A/B on one box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03stem}; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_fp16.py tests/test_gpu_benchmark_parity.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; tail -n 4 $O/pytest.log | cut -c1-200
for i in 1 2; do
for v in default stem_base stem_planes stem_fold; do
  L=""; [ $v != default ] && L=$R/sleap_amd/lib/alt/libsleap_amd_fp16_$v.so
  SLEAP_AMD_LIB_FP16=$L timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['network_ms_per_step'], j['roofline']['frac'], j['roofline']['frac_forward'])" | tee -a $O/ab.txt
  grep -E "^stem" $O/layers_${v}_$i.log | tee -a $O/ab.txt
done; done
