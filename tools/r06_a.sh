# round 6, first session: decoder probe, this box's baseline bench line and per-layer table (before any kernel change)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06a}; mkdir -p $O; cd $R
bash tools/decoder_probe.sh > $O/decoder_probe.txt 2>&1; cat $O/decoder_probe.txt
timeout 600 python bench.py --cpu-baseline-seconds 5 > $O/bench_line.json 2> $O/bench_err.log; python -c "
import json; j=json.loads(open('$O/bench_line.json').readline()); r=j['roofline']; print('BOX', j['value'], j['ms_per_step'], r['frac'], r['frac_step'], r['frac_forward'], r['frac_dense'], j['literal_split_8_per_gpu']['ms_per_step'], j['sustained']['value'])"
timeout 300 python bench.py --no-cpu-baseline --no-extras --layers --steps 20 > $O/bench_layers.json 2> $O/layers.log; grep -v amdgpu.ids $O/layers.log > $O/bench_layers.txt; cat $O/bench_layers.txt | cut -c1-120
