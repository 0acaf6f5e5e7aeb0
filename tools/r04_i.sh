# round 4, GPU call I: the whole suite after the distributed range agreement change + the RCCL path on one rank
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04i}; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 25 $O/pytest.log | cut -c1-300
timeout 600 python bench.py --force-dist --no-cpu-baseline --no-extras > $O/bench_force_dist.json 2> $O/bench_force_dist.err; cut -c1-200 $O/bench_force_dist.json; tail -n 3 $O/bench_force_dist.err
