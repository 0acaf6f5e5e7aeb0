# Round-3 profile set (run on the GPU box through gpurun): the default bench line, its rocprofv3 kernel trace, the three PMC
# passes (MFMA busy, FETCH_SIZE, WRITE_SIZE -- separate passes, kernel trace only), a kernel trace at 8 frames per GPU, the
# strong-scaling line and the other BASELINE configurations. Summaries are written next to the raw output; copy what is to be
# judged into profiles/.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03}
mkdir -p $O
cd $R
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.log
timeout 300 python bench.py --no-cpu-baseline --layers --steps 20 > $O/bench_layers.json 2> $O/layers.log
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python bench.py --no-cpu-baseline --no-extras > $O/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o run -- python bench.py --no-cpu-baseline --no-extras > $O/mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- python bench.py --no-cpu-baseline --no-extras > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o run -- python bench.py --no-cpu-baseline --no-extras > $O/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_b8 -o run -- python bench.py --no-cpu-baseline --batch 8 --steps 40 > $O/kt_b8.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --batch 8 --steps 40 > $O/bench_b8.json 2> $O/b8.log
timeout 300 python bench.py --no-cpu-baseline --global-batch 64 --steps 20 > $O/bench_g64.json 2> $O/g64.log
timeout 300 python bench.py --no-cpu-baseline --force-dist --steps 20 > $O/bench_forcedist.json 2> $O/fd.log
SA_FUSE_UPSAMPLE=0 timeout 300 python bench.py --no-cpu-baseline --no-extras --layers --steps 20 > $O/bench_layers_materialised.json 2> $O/layers_materialised.log
timeout 300 python bench.py --no-cpu-baseline --dtype bf16 --steps 20 > $O/bench_bf16.json 2> $O/bf16.log
timeout 300 python tools/bench_configs.py > $O/other_configs.md 2> $O/other_configs.err
for d in kt kt_b8; do
  db=$(find $O/$d -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/${d}_kernel_stats.md
done
f=$(find $O/mfma -name "*counter_collection.csv" | head -1); k=$(find $O/mfma -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/pmc_mfma.py $f $k > $O/pmc_mfma_util.md 2>&1
ff=$(find $O/fetch -name "*counter_collection.csv" | head -1); fw=$(find $O/write -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_traffic.py $ff $fw 0 $O/pmc_hbm_traffic.json > $O/pmc_hbm_traffic.md 2>&1
rm -rf $O/kt $O/kt_b8 $O/mfma $O/fetch $O/write
du -sh $O; ls $O
cut -c1-900 $O/bench_line.json
