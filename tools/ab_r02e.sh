cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02e; O=gpurun_out/r02e; L=$GRAFT_REPO_ROOT/sleap_amd/lib
for i in 1 2; do
for v in base stem1 mt4; do
  unset SLEAP_AMD_LIB_FP16 SA_CONV_MT4
  case $v in base) ;; stem1) export SLEAP_AMD_LIB_FP16=$L/alt_stem_1term.so;; mt4) export SA_CONV_MT4=256;; esac
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --parity-frames 8 --cpu-baseline-seconds 3 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['network_ms_per_step'], j.get('cpu_baseline',{}).get('parity_vs_oracle'))"
  grep "stem\|512->512\|768" $O/layers_${v}_$i.log | cut -c1-62 | tr '\n' ';'; echo
done; done
