# SQ counters of convpair64_kernel alone (tools/pair64_probe.py: 64 frames of 256 x 256, dense random data), three passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-p64pmc}; mkdir -p $O; cd $R
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/a -o run -- python tools/pair64_probe.py 64 > $O/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/b -o run -- python tools/pair64_probe.py 64 > $O/b.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC --output-format csv -d $O/c -o run -- python tools/pair64_probe.py 64 > $O/c.log 2>&1
python tools/pmc_sq.py $(find $O -name "*counter_collection.csv") --min-grid 1000 > $O/sq.md 2>&1; grep -i "pair64\|kernel |" $O/sq.md | cut -c1-400
tail -2 $O/a.log
