# stem16: triplet-table row stride 48 -> 40 / 36 entries = 28.0 -> 26.2 / 25.5 KB of LDS = 5 -> 6 workgroups per CU
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02y; O=gpurun_out/r02y
for i in 1 2 3; do
for v in base w6; do
  unset SLEAP_AMD_LIB_FP16
  case $v in base) ;; *) export SLEAP_AMD_LIB_FP16=$PWD/sleap_amd/lib/alt_$v.so;; esac
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['frac'])"
  grep "stem" $O/layers_${v}_$i.log | cut -c1-75
done; done
