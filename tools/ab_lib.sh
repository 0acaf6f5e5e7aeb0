VF=$GRAFT_REPO_ROOT/sleap_amd/lib/libsleap_amd_fp16_vf.so
for i in 1 2; do
for v in base vf; do
  if [ $v = vf ]; then export SLEAP_AMD_LIB_FP16=$VF; else unset SLEAP_AMD_LIB_FP16; fi
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline 2> gpurun_out/layers_$v.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['network_ms_per_step'])"
done; done
paste <(grep "ms " gpurun_out/layers_base.log | cut -c1-60) <(grep "ms " gpurun_out/layers_vf.log | cut -c45-60)
