# round 3: the checks the driver runs at round end -- the whole -m gpu suite, smoke(), the default bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03z}; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 6 $O/pytest.log | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.log; cut -c1-220 $O/bench_line.json; python -c "
import json; j=json.loads(open('$O/bench_line.json').readline()); print({k: j['roofline'][k] for k in ('frac','frac_forward','frac_materialised','frac_forward_materialised','frac_dense','traffic','network_ms_per_step')}); print(j['sustained']); print(j['literal_split_8_per_gpu']); print(j['cpu_baseline']['parity_vs_oracle'])"
