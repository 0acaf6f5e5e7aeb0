cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03e}; mkdir -p $O; cd $R
export SA_FUSE_UPSAMPLE=1
for i in 1 2; do
for v in base stag8 stag16 stag32 stag64; do
  if [ $v = base ]; then unset SLEAP_AMD_LIB_FP16; else export SLEAP_AMD_LIB_FP16=$R/sleap_amd/lib/libalt_$v.so; fi
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['network_ms_per_step'], j['roofline']['frac'], j['roofline']['frac_forward'])" | tee -a $O/ab.txt
  grep -E "mode2" $O/layers_${v}_$i.log | tee -a $O/ab.txt
done; done
