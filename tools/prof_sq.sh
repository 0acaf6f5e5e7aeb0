cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03sq}; mkdir -p $O; cd $R
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/a -o run -- python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $O/b -o run -- python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/b.log 2>&1
python tools/pmc_sq.py $O/a/run_counter_collection.csv $O/b/run_counter_collection.csv > $O/sq.md; cat $O/sq.md
