# round 6: the fused 32 -> 64 -> 64 block -- bitwise tests, the network tests that see the new plan, then A/B per-layer tables
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06b}; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_network.py -x -q -p no:cacheprovider -k "pair" > $O/pytest_pair.log 2>&1; tail -n 15 $O/pytest_pair.log | cut -c1-250
for v in 1 0; do
SA_FUSE_PAIRS64=$v timeout 300 python bench.py --no-cpu-baseline --no-extras --layers --steps 20 > $O/bench_layers_$v.json 2> $O/layers_$v.log; grep -v amdgpu.ids $O/layers_$v.log | cut -c1-110 > $O/bench_layers_$v.txt; head -8 $O/bench_layers_$v.txt; python -c "
import json; j=json.loads(open('$O/bench_layers_$v.json').readline()); print('fuse64=$v', j['value'], j['ms_per_step'], j['roofline']['frac_step'])"
done
