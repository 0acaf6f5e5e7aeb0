# round 5, session e: robustness of what stays switched off / new: the whole GPU suite with the persistent two-workgroup kernels ON
# (SA_CONV_PERS=1: every e2e parity test then runs through them), the RCCL path on one rank with the new line keys, a second bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05e}; mkdir -p $O; cd $R
SA_CONV_PERS=1 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_pers1.log 2>&1; echo "pytest rc $?" >> $O/pytest_pers1.log; tail -n 5 $O/pytest_pers1.log | cut -c1-300
timeout 600 python bench.py --force-dist --no-cpu-baseline --steps 10 > $O/force_dist.json 2> $O/force_dist.err; python -c "
import json; j=json.loads(open('$O/force_dist.json').readline()); print(j['value'], j['config']['collective_backend'], j['value_weak_64_per_gpu'], j['value_strong_global_batch_64'], j['configs3_global_batch_64'], j['roofline']['frac_step'], j['roofline']['plan'][:60])"
timeout 600 python bench.py --no-extras --no-cpu-baseline --global-batch 64 --steps 10 > $O/gb64.json 2> $O/gb64.err; python -c "
import json; j=json.loads(open('$O/gb64.json').readline()); print(j['value'], j['scaling'], j['value_weak_64_per_gpu'], j['value_strong_global_batch_64'], j['roofline']['plan'][:40])"
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.log; python -c "
import json; j=json.loads(open('$O/bench_line.json').readline()); print(j['value'], j['ms_per_step'], {k: j['roofline'][k] for k in ('frac','frac_step','frac_forward','frac_dense')}, j['literal_split_8_per_gpu'])"
