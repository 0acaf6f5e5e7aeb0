# round 4, GPU call G: the parity tests with their printed numbers (-s), then one default bench line (another box of the pool)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04g}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_config_parity.py tests/test_gpu_benchmark_parity.py tests/test_gpu_postproc_mouse24.py -m gpu -q -s -p no:cacheprovider > $O/parity.log 2>&1; grep -E "configs\[|two-stack|variant|c[01]_|hg_|passed|failed" $O/parity.log | cut -c1-330
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.log; python -c "
import json; j=json.loads(open('$O/bench_line.json').readline()); r=j['roofline']; print(j['value'], j['ms_per_step'], {k: r[k] for k in ('frac','frac_forward','frac_dense','network_ms_per_step')}, j['sustained'], j['cpu_baseline']['value'], j['cpu_baseline']['parity_vs_oracle']['tolerance_met'])"
