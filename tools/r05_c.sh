# round 5, session c: persistent-kernel tests with poisoned outputs; sched_group_barrier builds (all kernels / all but the fused-head
# kernels) against the default, three alternations; SQ counters of the plain 3x3 layers, layer by layer (tools/pmc_layers.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05c}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_fp16.py tests/test_gpu_postproc.py tests/test_abi.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 8 $O/pytest.log | cut -c1-300
run() {  # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extras --layers --steps 30 > $O/bench_$n.json 2> $O/layers_$n.txt
  python -c "
import json; j=json.loads(open('$O/bench_$n.json').readline()); print('$n', j['value'], j['ms_per_step'], {k: j['roofline'][k] for k in ('frac','frac_step','frac_forward','network_ms_per_step')})"
}
S2=SLEAP_AMD_LIB_FP16=$R/sleap_amd/lib/libsleap_amd_fp16_sgb2.so
S3=SLEAP_AMD_LIB_FP16=$R/sleap_amd/lib/libsleap_amd_fp16_sgb3.so
for i in 1 2 3; do
  run base_$i A=1
  run sgb2_$i $S2
  run sgb3_$i $S3
done
for n in base_1 sgb2_1 sgb3_1 base_2 sgb2_2 sgb3_2 base_3 sgb2_3 sgb3_3; do grep -v amdgpu $O/layers_$n.txt | awk '{print $(NF-3)}' > $O/col_$n.txt; done
grep -v amdgpu $O/layers_base_1.txt | awk '{$NF="";$(NF-1)="";$(NF-2)="";$(NF-3)="";print}' > $O/col_names.txt
echo "layer | base sgb2 sgb3 (x3)"; paste $O/col_names.txt $O/col_base_1.txt $O/col_sgb2_1.txt $O/col_sgb3_1.txt $O/col_base_2.txt $O/col_sgb2_2.txt $O/col_sgb3_2.txt $O/col_base_3.txt $O/col_sgb2_3.txt $O/col_sgb3_3.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pa -o run -- python tools/pmc_layers.py run 64 > $O/pa.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pb -o run -- python tools/pmc_layers.py run 64 > $O/pb.log 2>&1
tail -3 $O/pa.log $O/pb.log | cut -c1-200
fa=$(find $O/pa -name "*counter_collection.csv" | head -1); fb=$(find $O/pb -name "*counter_collection.csv" | head -1)
python tools/pmc_layers.py report $fa $fb > $O/pmc_dominant_per_layer.md 2>&1; cat $O/pmc_dominant_per_layer.md | cut -c1-260
rm -rf $O/pa $O/pb
