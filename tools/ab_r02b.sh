# round-2 A/B session (one box): planar layout per layer shape, convpair v2, stem16 XCD order / half swizzle
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02b; O=gpurun_out/r02b; L=$GRAFT_REPO_ROOT/sleap_amd/lib
timeout 300 python tools/planar_ab.py 64 20 > $O/planar_ab.md 2> $O/planar_ab.err
echo "== pair v2 tests"; SA_PAIR_V2=1 timeout 200 python -m pytest tests/test_gpu_network.py tests/test_gpu_fp16.py -q -x 2>&1 | tail -2
for v in alt_stem_xcd.so alt_stem_swz.so; do echo "== tests $v"; SLEAP_AMD_LIB_FP16=$L/$v timeout 200 python -m pytest tests/test_gpu_network.py tests/test_gpu_fp16.py -q -x 2>&1 | tail -2; done
for i in 1 2; do
for v in base pairv2 alt_stem_xcd.so alt_stem_swz.so; do
  unset SLEAP_AMD_LIB_FP16 SA_PAIR_V2
  case $v in base) ;; pairv2) export SA_PAIR_V2=1;; *) export SLEAP_AMD_LIB_FP16=$L/$v;; esac
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['frac'])"
  grep "stem\|pair" $O/layers_${v}_$i.log | cut -c1-62
done; done
cat $O/planar_ab.md
