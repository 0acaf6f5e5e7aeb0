# round 4, GPU call B: the -m gpu suite, the ResNet-50 network on planes vs NHWC (tools/net_profile.py), SQ counters of the bench
# kernels on the current code (tools/prof_sq.sh), one env A/B (the 128 x 128 decoder stage's upsampling fused as well).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; L=$R/sleap_amd/lib; O=$R/gpurun_out/${1:-r04b}; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 30 $O/pytest.log | cut -c1-300
for lay in planes16 nhwc planes16 nhwc; do
  [ $lay = nhwc ] && export SA_LAYOUT=nhwc || unset SA_LAYOUT
  timeout 300 python tools/net_profile.py resnet50 1024 16 0.012 >> $O/resnet50_$lay.txt 2>&1; tail -n 10 $O/resnet50_$lay.txt
done
unset SA_LAYOUT
for i in 1 2; do
for v in base up128; do
  unset SA_FUSE_UPSAMPLE_MAX_COUT
  [ $v = up128 ] && export SA_FUSE_UPSAMPLE_MAX_COUT=128
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); r=j['roofline']; print('$v', j['value'], j['ms_per_step'], r['network_ms_per_step'], r['frac'], r['frac_forward'])" | tee -a $O/ab.txt
done; done
unset SA_FUSE_UPSAMPLE_MAX_COUT
bash tools/prof_sq.sh ${1:-r04b}/sq > $O/prof_sq.log 2>&1; tail -n 25 $O/prof_sq.log | cut -c1-260
