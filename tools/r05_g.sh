# does a smaller step (sub-batches whose layer-to-layer tensors fit the 256 MB Infinity Cache) run at a higher rate per frame?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05g}; mkdir -p $O; cd $R
for b in 64 32 16 24 12 64 32 16; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --batch $b --steps $((1280 / b)) > $O/b$b.json 2> $O/b$b.err
  python -c "
import json; j=json.loads(open('$O/b$b.json').readline()); print('batch', $b, j['value'], 'frames/s', j['ms_per_step'], 'ms/step =', round(j['ms_per_step'] / $b * 64, 3), 'ms per 64 frames; network', round(j['roofline']['network_ms_per_step'] / $b * 64, 3))"
done
