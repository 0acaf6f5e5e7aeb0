# round 3: A/B of the fused (in-LDS, packed-fp16) upsampling against the materialised one, alternating runs on ONE box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03d}; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_network.py -m gpu -q -s -p no:cacheprovider -k "upsampl or layout" > $O/pytest.log 2>&1; tail -n 8 $O/pytest.log | cut -c1-200
for i in 1 2 3; do
for v in 0 1; do
  SA_FUSE_UPSAMPLE=$v timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('fuse=$v', j['value'], j['ms_per_step'], j['roofline']['network_ms_per_step'], j['roofline']['frac'], j['roofline']['frac_forward'])" | tee -a $O/ab.txt
done; done
grep -E "mode1|mode2|up " $O/layers_0_3.log | tee -a $O/ab.txt; echo ---- | tee -a $O/ab.txt; grep -E "mode1|mode2|mode3|up " $O/layers_1_3.log | tee -a $O/ab.txt
