# round 5, session a: the GPU suite on the new code (PERS kernels, asserting fixture tests), then PERS on / off alternating on the
# per-layer table, then the s_memtime stall attribution of the plain 3x3 layers (instrumented build) with PERS on and off
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05a}; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 15 $O/pytest.log | cut -c1-300
for i in 1 2; do
  for v in 1 0; do
    SA_CONV_PERS=$v timeout 300 python bench.py --no-cpu-baseline --no-extras --layers --steps 20 > $O/bench_pers${v}_$i.json 2> $O/layers_pers${v}_$i.txt
    python -c "
import json; j=json.loads(open('$O/bench_pers${v}_$i.json').readline()); print('PERS=$v run $i', j['value'], j['ms_per_step'], {k: j['roofline'][k] for k in ('frac','frac_step','frac_forward','network_ms_per_step')})"
  done
done
paste $O/layers_pers1_1.txt $O/layers_pers0_1.txt | grep -v amdgpu | awk '{printf "%-44s %8s %8s\n", $1" "$2" "$3" "$4, $(NF/2-3), $(NF-3)}' | head -30
SLEAP_AMD_LIB_FP16=$R/sleap_amd/lib/libsleap_amd_fp16_stamp.so timeout 300 python tools/stall_probe.py 64 > $O/stall_pers1.md 2> $O/stall_pers1.err; cat $O/stall_pers1.md | cut -c1-220
SA_CONV_PERS=0 SLEAP_AMD_LIB_FP16=$R/sleap_amd/lib/libsleap_amd_fp16_stamp.so timeout 300 python tools/stall_probe.py 64 > $O/stall_pers0.md 2> $O/stall_pers0.err; cat $O/stall_pers0.md | cut -c1-220
tail -3 $O/stall_pers1.err
