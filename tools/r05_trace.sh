# round 5: the bench line + kernel trace + MFMA / clock pass of one session on whatever box the pool hands out (the box spread of
# the fraction of peak is the clock the box sustains: profiles/r05_ab_session.md section 3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05t}; mkdir -p $O; cd $R
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.log; python -c "
import json; j=json.loads(open('$O/bench_line.json').readline()); print(j['value'], j['ms_per_step'], {k: j['roofline'][k] for k in ('frac','frac_step','frac_forward','frac_dense','frac_forward_dense')}, j['literal_split_8_per_gpu'], j['cpu_baseline']['value'])"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python bench.py --no-cpu-baseline --no-extras > $O/kt.log 2>&1
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/kt_kernel_stats.md
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o run -- python bench.py --no-cpu-baseline --no-extras > $O/mfma.log 2>&1
f=$(find $O/mfma -name "*counter_collection.csv" | head -1); k=$(find $O/mfma -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/pmc_mfma.py $f $k > $O/pmc_mfma_util.md 2>&1
rm -rf $O/kt $O/mfma
python - <<PY
import re
conv=up=0.0; n=None
for l in open('$O/kt_kernel_stats.md'):
    m=re.match(r'\| \`([^\`]*)\` \| (\d+) \| ([\d.]+) \|',l)
    if not m: continue
    nm=m.group(1)
    if 'stem16_gray' in nm: n=int(m.group(2))
    if any(k in nm for k in ('conv3x3_dma','stem16','convpair')): conv+=float(m.group(3))
    if 'upsample2x' in nm: up+=float(m.group(3))
fl=99.35e9*64
print('forward passes', n, 'conv ms/step %.3f -> %.4f ; with upsampling %.3f -> %.4f of 2.5 PFLOP/s' % (conv/n, fl/(conv/n*1e-3)/2.5e15, (conv+up)/n, fl/((conv+up)/n*1e-3)/2.5e15))
PY
tail -2 $O/pmc_mfma_util.md | cut -c1-200
