# round 6 (the round-5 script with this round's additions): the checks the driver runs at round end -- the whole -m gpu suite, smoke(), the default bench line; with a second argument
# the profile set of the same code: kernel trace, MFMA / FETCH / WRITE counter passes (separate runs), the MFMA + clock pass on DENSE
# (random-init) activations, SQ counters, the per-layer SQ table of the dominant kernel (tools/pmc_layers.py), per-layer times, the
# 8-frame kernel trace, the COMPLETE ResNet-50 per-launch table + its kernel trace and counter passes, the other BASELINE configs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06z}; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 6 $O/pytest.log | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.log; cut -c1-220 $O/bench_line.json; python -c "
import json; j=json.loads(open('$O/bench_line.json').readline()); print({k: j['roofline'][k] for k in ('frac','frac_step','frac_forward','frac_materialised','frac_forward_materialised','frac_dense','frac_forward_dense','traffic','network_ms_per_step')}); print(j['sustained']); print(j['literal_split_8_per_gpu']); print(j['configs3_global_batch_64'], j['value_weak_64_per_gpu'], j['value_strong_global_batch_64']); print(j['cpu_baseline']); print(j['roofline_postproc'])"
if [ -n "$2" ]; then
timeout 300 python bench.py --no-cpu-baseline --no-extras --layers --steps 20 > $O/bench_layers.json 2> $O/layers.log; grep -v amdgpu.ids $O/layers.log > $O/bench_layers.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python bench.py --no-cpu-baseline --no-extras > $O/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o run -- python bench.py --no-cpu-baseline --no-extras > $O/mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfmad -o run -- python bench.py --no-cpu-baseline --no-extras --random-init > $O/mfmad.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- python bench.py --no-cpu-baseline --no-extras > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o run -- python bench.py --no-cpu-baseline --no-extras > $O/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_b8 -o run -- python bench.py --no-cpu-baseline --no-extras --batch 8 --steps 40 > $O/kt_b8.log 2>&1
for d in kt kt_b8; do db=$(find $O/$d -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/${d}_kernel_stats.md; done
f=$(find $O/mfma -name "*counter_collection.csv" | head -1); k=$(find $O/mfma -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/pmc_mfma.py $f $k > $O/pmc_mfma_util.md 2>&1
f=$(find $O/mfmad -name "*counter_collection.csv" | head -1); k=$(find $O/mfmad -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/pmc_mfma.py $f $k > $O/pmc_mfma_util_dense.md 2>&1
ff=$(find $O/fetch -name "*counter_collection.csv" | head -1); fw=$(find $O/write -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_traffic.py $ff $fw 0 $O/pmc_hbm_traffic.json > $O/pmc_hbm_traffic.md 2>&1
rm -rf $O/kt $O/kt_b8 $O/mfma $O/mfmad $O/fetch $O/write
bash tools/prof_sq.sh ${1:-r06z}/sq > $O/prof_sq.log 2>&1; cp $O/sq/sq.md $O/pmc_sq_counters.md; rm -rf $O/sq/a $O/sq/b
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pa -o run -- python tools/pmc_layers.py run 64 > $O/pa.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pb -o run -- python tools/pmc_layers.py run 64 > $O/pb.log 2>&1
fa=$(find $O/pa -name "*counter_collection.csv" | head -1); fb=$(find $O/pb -name "*counter_collection.csv" | head -1)
python tools/pmc_layers.py report $fa $fb > $O/pmc_dominant_per_layer.md 2>&1; rm -rf $O/pa $O/pb
# ResNet-50 (configs[4] architecture): EVERY launch, kernel trace, FETCH / WRITE / MFMA passes
timeout 300 python tools/net_profile.py resnet50 1024 16 2>&1 | grep -v amdgpu.ids > $O/resnet50_per_launch.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/rkt -o run -- python tools/net_profile.py resnet50 1024 16 1.0 > $O/rkt.log 2>&1
db=$(find $O/rkt -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/resnet50_kernel_stats.md
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/rfetch -o run -- python tools/net_profile.py resnet50 1024 16 1.0 > $O/rfetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/rwrite -o run -- python tools/net_profile.py resnet50 1024 16 1.0 > $O/rwrite.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/rmfma -o run -- python tools/net_profile.py resnet50 1024 16 1.0 > $O/rmfma.log 2>&1
ff=$(find $O/rfetch -name "*counter_collection.csv" | head -1); fw=$(find $O/rwrite -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_traffic_generic.py $ff $fw > $O/resnet50_pmc_hbm_traffic.md 2>&1
f=$(find $O/rmfma -name "*counter_collection.csv" | head -1); k=$(find $O/rmfma -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/pmc_mfma.py $f $k > $O/resnet50_pmc_mfma_util.md 2>&1
rm -rf $O/rkt $O/rfetch $O/rwrite $O/rmfma
timeout 300 python tools/net_profile.py hourglass 1024 16 2>&1 | grep -v amdgpu.ids > $O/hourglass_per_launch.txt
# round 6: the fused 32 -> 64 -> 64 block alone -- time on dense data, s_memtime segment sums of the stamp build (if present), SQ counters
timeout 120 python tools/pair64_probe.py 64 2>/dev/null > $O/pair64_probe.txt
[ -f sleap_amd/lib/libsleap_amd_fp16_p64stamp.so ] && SLEAP_AMD_LIB_FP16=sleap_amd/lib/libsleap_amd_fp16_p64stamp.so timeout 120 python tools/pair64_probe.py 64 2>/dev/null >> $O/pair64_probe.txt
bash tools/pair64_pmc.sh ${1:-r06z}/p64pmc > /dev/null 2>&1; python tools/pmc_sq.py $(find $O/p64pmc -name "*counter_collection.csv") --min-grid 1000 2>/dev/null | grep -i "pair64\|kernel |" > $O/pair64_sq.md; rm -rf $O/p64pmc/a $O/p64pmc/b $O/p64pmc/c
SA_FUSE_PAIRS64=0 timeout 300 python bench.py --no-cpu-baseline --no-extras --layers --steps 20 > $O/bench_layers_nofuse64.json 2> $O/layers_nofuse64.log; grep -v amdgpu.ids $O/layers_nofuse64.log > $O/bench_layers_nofuse64.txt
timeout 400 python tools/bench_configs.py 10 > $O/other_configs.md 2> $O/other_configs.err; cat $O/other_configs.md | cut -c1-200
ls $O; head -30 $O/kt_kernel_stats.md | cut -c1-200
fi
