# round 6: the per-layer rule of the upsampling fusion (expansion in LDS inside the consuming convolution) re-measured after the
# expansion's bank-conflict fix: SA_FUSE_UPSAMPLE_MAX_COUT = 64 (the 256x256 stage only: default), 128 (+ the 128x128 stage), 256 (+ 64x64)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-upsab}; mkdir -p $O; cd $R
for i in 1 2 3; do for mc in 64 128 256; do
  SA_FUSE_UPSAMPLE_MAX_COUT=$mc timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 30 --layers 2> $O/layers_${mc}_$i.log | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); r=j['roofline']; print('max_cout $mc:', j['value'], j['ms_per_step'], 'network', r['network_ms_per_step'], 'frac_step', r['frac_step'], j['config']['result_digest'])" | tee -a $O/ab.txt
done; done
for mc in 64 128 256; do echo "== $mc"; grep -E "up |mode1|mode2" $O/layers_${mc}_2.log | cut -c1-80; done | tee -a $O/ab.txt
