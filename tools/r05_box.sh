# one default bench line on whatever box the pool hands out (box spread of the same binaries)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05box}; mkdir -p $O; cd $R
timeout 600 python bench.py --cpu-baseline-seconds 5 > $O/bench_line.json 2> $O/bench_err.log; python -c "
import json; j=json.loads(open('$O/bench_line.json').readline()); r=j['roofline']; print('BOX', j['value'], j['ms_per_step'], r['frac'], r['frac_step'], r['frac_forward'], r['frac_dense'], j['literal_split_8_per_gpu']['ms_per_step'], j['sustained']['value'])"
