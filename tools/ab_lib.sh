# A/B of two builds of the fp16 library on ONE box (box-to-box variance is +-3 %): bash tools/ab_lib.sh <other libsleap_amd_fp16.so>
# The alternative file is loaded through SLEAP_AMD_LIB_FP16 (sleap_amd/_lib.py). Prints frames/s of alternating runs and the
# per-layer tables side by side.
ALT=${1:?path of the alternative library (inside the repo, so that it travels to the GPU box)}
for i in 1 2; do
for v in base alt; do
  if [ $v = alt ]; then export SLEAP_AMD_LIB_FP16=$ALT; else unset SLEAP_AMD_LIB_FP16; fi
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline 2> gpurun_out/layers_$v.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['network_ms_per_step'])"
done; done
paste <(grep "ms " gpurun_out/layers_base.log | cut -c1-60) <(grep "ms " gpurun_out/layers_alt.log | cut -c45-60)
