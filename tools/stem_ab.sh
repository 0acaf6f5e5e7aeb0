# round 6: the two stem16 leftovers of DESIGN section 7.1 (ds_read2_b64 operand fold, v_max_f32_dpp) A/B'd on one box.
#   bash tools/stem_ab.sh <out-name> <alt1.so> <alt2.so> ...   (alternates made by tools/build_alt.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; L=$R/sleap_amd/lib; O=$R/gpurun_out/${1:-stemab}; shift; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_fp16.py tests/test_gpu_benchmark_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 4 | cut -c1-200 | tee $O/tests.txt
for i in 1 2 3; do
for v in base "$@"; do
  if [ $v = base ]; then unset SLEAP_AMD_LIB_FP16; else export SLEAP_AMD_LIB_FP16=$L/$v; fi
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); r=j['roofline']; print('$v', j['value'], j['ms_per_step'], r['network_ms_per_step'], r['frac'], r['frac_forward'], j['config']['result_digest'])" | tee -a $O/ab.txt
  grep "stem" $O/layers_${v}_$i.log | cut -c1-90 | tee -a $O/ab.txt
done; done
