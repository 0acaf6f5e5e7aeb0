cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03s}; mkdir -p $O; cd $R
for i in 1 2; do for v in base spread lt4 lt6 lt7; do
  if [ $v = base ]; then unset SLEAP_AMD_LIB_FP16; else export SLEAP_AMD_LIB_FP16=$R/sleap_amd/lib/libalt_$v.so; fi
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['network_ms_per_step'], j['roofline']['frac'], j['roofline']['frac_forward'])" | tee -a $O/ab.txt
  grep -E "mode2" $O/layers_${v}_$i.log | tee -a $O/ab.txt
done; done
SA_FUSE_UPSAMPLE=1 SLEAP_AMD_LIB_FP16=$R/sleap_amd/lib/libalt_spread.so timeout 300 python -m pytest tests/test_gpu_network.py -m gpu -q -p no:cacheprovider -k "upsampl" 2>&1 | tail -n 2
