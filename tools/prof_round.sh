set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/v10
mkdir -p $O
cd $R
timeout 300 python bench.py > $O/bench_line.json 2> $O/bench_err.log
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python bench.py --no-cpu-baseline > $O/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o run -- python bench.py --no-cpu-baseline > $O/mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- python bench.py --no-cpu-baseline > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o run -- python bench.py --no-cpu-baseline > $O/write.log 2>&1
find $O -type f | head -40
du -sh $O
cat $O/bench_line.json | cut -c1-600
