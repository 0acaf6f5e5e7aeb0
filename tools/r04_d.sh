# round 4, GPU call D: fused ResNet bottleneck tails -- correctness, then the network with and without them (one box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04d}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_backbones.py -m gpu -q -x -p no:cacheprovider -k "resnet" 2>&1 | tail -n 25 | cut -c1-300
for v in 1 0 1 0; do
  export SA_FUSE_BNECK=$v
  timeout 300 python tools/net_profile.py resnet50 1024 16 0.012 > $O/rn_fuse$v.txt 2>&1; grep -E "bneck|^conv1x1s1|^conv3x3 |^conv |^total" $O/rn_fuse$v.txt | cut -c1-120
done
unset SA_FUSE_BNECK
for i in 1 2; do for v in 4 99; do
  export SA_CONV_NT_MAX_CHUNKS=$v
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2> $O/layers_nt${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); r=j['roofline']; print('nt_max_chunks $v', j['value'], j['ms_per_step'], r['network_ms_per_step'], r['frac'], r['frac_forward'])" | tee -a $O/ab.txt
done; done
grep "192->64" $O/layers_nt*_2.log
