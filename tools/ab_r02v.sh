# the weights' copies at a tap of their own: few-chunk layers (inputs stay right after the barrier) and many-chunk layers
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02v; O=gpurun_out/r02v
for i in 1 2; do
for v in base few0 few2 few4; do
  unset SLEAP_AMD_LIB_FP16
  case $v in base) ;; *) export SLEAP_AMD_LIB_FP16=$PWD/sleap_amd/lib/alt_$v.so;; esac
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['frac'])"
done; done
paste <(grep "ms " $O/layers_base_2.log | cut -c1-60) <(grep "ms " $O/layers_few0_2.log | cut -c45-60) <(grep "ms " $O/layers_few2_2.log | cut -c45-60) <(grep "ms " $O/layers_few4_2.log | cut -c45-60)
