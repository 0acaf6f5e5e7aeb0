cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02i; O=gpurun_out/r02i; L=$GRAFT_REPO_ROOT/sleap_amd/lib
for i in 1 2 3; do
for v in occ4 new; do
  unset SLEAP_AMD_LIB_FP16
  case $v in new) ;; occ4) export SLEAP_AMD_LIB_FP16=$L/alt_stem_occ4.so;; esac
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['frac'])"
  grep "stem" $O/layers_${v}_$i.log | cut -c1-62
done; done
