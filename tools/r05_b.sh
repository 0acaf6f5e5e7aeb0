# round 5, session b: the suite again (fixture test bound), then the persistent two-workgroup kernels in their second form (cross-tile
# prefetch after the chunk loop) -- off / on / few-chunk layers only -- and the sched_group_barrier builds, per-layer tables
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05b}; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 12 $O/pytest.log | cut -c1-300
run() {  # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extras --layers --steps 20 > $O/bench_$n.json 2> $O/layers_$n.txt
  python -c "
import json; j=json.loads(open('$O/bench_$n.json').readline()); print('$n', j['value'], j['ms_per_step'], {k: j['roofline'][k] for k in ('frac','frac_step','frac_forward','network_ms_per_step')})"
}
S1=SLEAP_AMD_LIB_FP16=$R/sleap_amd/lib/libsleap_amd_fp16_sgb1.so
S2=SLEAP_AMD_LIB_FP16=$R/sleap_amd/lib/libsleap_amd_fp16_sgb2.so
for i in 1 2; do
  run pers0_$i SA_CONV_PERS=0
  run pers1_$i SA_CONV_PERS=1
  run pers2_$i SA_CONV_PERS=2
  run sgb1_pers1_$i SA_CONV_PERS=1 $S1
  run sgb2_pers0_$i SA_CONV_PERS=0 $S2
done
for n in pers0_1 pers1_1 pers2_1 sgb1_pers1_1 sgb2_pers0_1 pers0_2 pers1_2 sgb2_pers0_2; do grep -v amdgpu $O/layers_$n.txt | awk '{print $(NF-3)}' > $O/col_$n.txt; done
grep -v amdgpu $O/layers_pers0_1.txt | awk '{$NF="";$(NF-1)="";$(NF-2)="";$(NF-3)="";print}' > $O/col_names.txt
echo "layer | pers0 pers1 pers2 sgb1+pers1 sgb2+pers0 | pers0' pers1' sgb2+pers0'"; paste $O/col_names.txt $O/col_pers0_1.txt $O/col_pers1_1.txt $O/col_pers2_1.txt $O/col_sgb1_pers1_1.txt $O/col_sgb2_pers0_1.txt $O/col_pers0_2.txt $O/col_pers1_2.txt $O/col_sgb2_pers0_2.txt
