# round 4, GPU call E: the whole suite (fused bottleneck tails included), then the tap GEMM's LDS stage count on the ResNet network
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04e}; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 25 $O/pytest.log | cut -c1-300
for ns in 2 3 4 2 3 4; do
  export SA_TAP_STAGES=$ns
  timeout 300 python tools/net_profile.py resnet50 1024 16 0.012 > $O/rn_ns$ns.txt 2>&1
  echo "stages $ns: $(grep -E '^conv1x1s1|^conv1x1s2|^convT4|^total' $O/rn_ns$ns.txt | cut -c1-62 | tr '\n' '|')" | tee -a $O/sweep_ns.txt
done
for ns in 3 4; do
  export SA_TAP_STAGES=$ns
  timeout 600 python -m pytest tests/test_gpu_backbones.py -m gpu -q -k "resnet or convt or conv1x1 or stem_block" -p no:cacheprovider 2>&1 | tail -n 3 | cut -c1-200
done
unset SA_TAP_STAGES
grep -E "conv1x1s[12] " $O/rn_ns2.txt | cut -c1-100 | head -40 > $O/per_layer_ns2.txt; grep -E "conv1x1s[12] " $O/rn_ns4.txt | cut -c1-100 | head -40 > $O/per_layer_ns4.txt
paste $O/per_layer_ns2.txt $O/per_layer_ns4.txt | cut -c1-70,100-175 | head -30
