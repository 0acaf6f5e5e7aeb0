# A/B of several builds of the fp16 library on ONE box: bash tools/ab_multi.sh <lib1.so> <lib2.so> ...   (files in sleap_amd/lib,
# made by tools/build_alt.py). The default build's parity tests first (AB_TESTS overrides the selection), then alternating
# bench runs: frames/s, step and network ms, conv-family fractions; the per-layer tables go to gpurun_out/ab/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; L=$R/sleap_amd/lib; O=$R/gpurun_out/${AB_OUT:-ab}; mkdir -p $O; cd $R
timeout 900 python -m pytest ${AB_TESTS:-tests/test_gpu_network.py tests/test_gpu_fp16.py} -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 4 | cut -c1-200
for i in 1 2 ${AB_ROUNDS:-}; do
for v in base "$@"; do
  if [ $v = base ]; then unset SLEAP_AMD_LIB_FP16; else export SLEAP_AMD_LIB_FP16=$L/$v; fi
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline --no-extras ${AB_BENCH_ARGS:-} 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); r=j['roofline']; print('$v', j['value'], j['ms_per_step'], r['network_ms_per_step'], r['frac'], r['frac_forward'])" | tee -a $O/ab.txt
done; done
unset SLEAP_AMD_LIB_FP16
for v in base "$@"; do echo "== $v"; grep " ms " $O/layers_${v}_2.log | cut -c1-75; done | tee -a $O/ab.txt
