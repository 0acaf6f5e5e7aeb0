# round 4, GPU call A: the whole -m gpu suite on the new code (stem16 fragment reuse, pow2 tile decode, tap GEMM on planes, range
# scan kernel, mouse24 post-processing parity), then one-box A/Bs: stem16 reuse on / off, tile decode by shifts on / off, the
# ResNet-50 network on planes vs NHWC (tools/net_profile.py).   bash tools/r04_a.sh <out name>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; L=$R/sleap_amd/lib; O=$R/gpurun_out/${1:-r04a}; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 40 $O/pytest.log | cut -c1-300
for i in 1 2; do
for v in base stem_old decode_div; do
  unset SLEAP_AMD_LIB_FP16 SA_CONV_POW2_DECODE
  [ $v = stem_old ] && export SLEAP_AMD_LIB_FP16=$L/alt_stem_old.so
  [ $v = decode_div ] && export SA_CONV_POW2_DECODE=0
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); r=j['roofline']; print('$v', j['value'], j['ms_per_step'], r['network_ms_per_step'], r['frac'], r['frac_forward'])" | tee -a $O/ab.txt
done; done
unset SLEAP_AMD_LIB_FP16 SA_CONV_POW2_DECODE
for v in base stem_old decode_div; do echo "== $v"; grep " ms " $O/layers_${v}_2.log | cut -c1-75; done >> $O/ab.txt
for lay in planes16 nhwc; do
  [ $lay = nhwc ] && export SA_LAYOUT=nhwc || unset SA_LAYOUT
  timeout 300 python tools/net_profile.py resnet50 1024 16 0.012 > $O/resnet50_$lay.txt 2>&1; tail -n 12 $O/resnet50_$lay.txt
done
unset SA_LAYOUT
