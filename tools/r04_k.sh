# round 4, GPU call K: heads fused behind the extended epilogue (SA_FUSE_EXT_HEADS 1 / 0) + the cout-tile-walking 1x1 kernel, whole suite
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04k}; mkdir -p $O; cd $R
if [ "$2" = "all" ]; then SEL="tests"; else SEL="tests/test_gpu_backbones.py tests/test_gpu_network.py tests/test_abi.py tests/test_gpu_config_parity.py"; fi
timeout 1500 python -m pytest $SEL -m gpu -q --maxfail=10 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 8 $O/pytest.log | cut -c1-300
for i in 1 2; do for v in 1 0; do
  echo "== SA_FUSE_EXT_HEADS=$v" >> $O/resnet.txt
  SA_FUSE_EXT_HEADS=$v timeout 300 python tools/net_profile.py resnet50 1024 16 0 2>/dev/null | grep -E "head|^total|64->64 @256 mode0|64->64 @128 mode0|^conv3x3 +n" >> $O/resnet.txt
done; done
cat $O/resnet.txt | cut -c1-140
