# A/B of several builds of the fp16 library on ONE box: bash tools/ab_multi.sh <lib1.so> <lib2.so> ...
# For each: the network parity tests with that library, then alternating bench runs (per-layer stem / pair times, frames/s).
L=$GRAFT_REPO_ROOT/sleap_amd/lib
for v in "$@"; do
  echo "== tests $v"; SLEAP_AMD_LIB_FP16=$L/$v timeout 200 python -m pytest tests/test_gpu_network.py tests/test_gpu_fp16.py -q -x 2>&1 | tail -1
done
for i in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then unset SLEAP_AMD_LIB_FP16; else export SLEAP_AMD_LIB_FP16=$L/$v; fi
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline 2> gpurun_out/layers_x.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'])"
  grep "stem\|pair" gpurun_out/layers_x.log | cut -c1-62
done; done
