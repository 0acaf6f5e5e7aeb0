# upsampling fused into the decoder convs (UPS DMA kernels on planes) vs the materialised upsampling
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02u; O=gpurun_out/r02u
timeout 900 python -m pytest tests/test_gpu_network.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
for v in 0 1; do
  SA_FUSE_UPSAMPLE=$v timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('fuse=$v', j['value'], j['ms_per_step'], j['roofline']['frac'], j.get('parity_vs_oracle',{}).get('tolerance_met'))"
done; done
grep -E "conv3x3|up|total" $O/layers_0_3.log | cut -c1-110 | head -30
echo ---; grep -E "conv3x3|up|total" $O/layers_1_3.log | cut -c1-110 | head -30
