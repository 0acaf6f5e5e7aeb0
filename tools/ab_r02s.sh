# where in a chunk the next chunk's copies are issued: right after the barrier (base) or after tap t of the MFMA sequence
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02s; O=gpurun_out/r02s
for i in 1 2; do
for v in base 4 6 7 8; do
  if [ $v = base ]; then unset SLEAP_AMD_LIB_FP16; else export SLEAP_AMD_LIB_FP16=$PWD/sleap_amd/lib/alt_issue$v.so; fi
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('issue=$v', j['value'], j['ms_per_step'], j['roofline']['frac'])"
done; done
paste <(grep "ms " $O/layers_base_2.log | cut -c1-60) <(grep "ms " $O/layers_6_2.log | cut -c45-60) <(grep "ms " $O/layers_7_2.log | cut -c45-60) <(grep "ms " $O/layers_8_2.log | cut -c45-60) <(grep "ms " $O/layers_4_2.log | cut -c45-60)
