cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02h; O=gpurun_out/r02h; L=$GRAFT_REPO_ROOT/sleap_amd/lib
for i in 1 2; do
for v in base alt_nt.so; do
  unset SLEAP_AMD_LIB_FP16
  case $v in base) ;; *) export SLEAP_AMD_LIB_FP16=$L/$v;; esac
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['network_ms_per_step'])"
done; done
paste <(grep "ms " $O/layers_base_2.log | cut -c1-60) <(grep "ms " $O/layers_alt_nt.so_2.log | cut -c45-60)
