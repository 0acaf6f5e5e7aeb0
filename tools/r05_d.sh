# round 5, session d: the tiled 3x3 / s2 max pool (tests + the ResNet-50 per-launch table with it on / off), the persistent tests after
# the grid fix, the fixture tests with their printed counts
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05d}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_backbones.py tests/test_gpu_persistent.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 6 $O/pytest.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_fp16.py tests/test_gpu_config_parity.py -m gpu -q -p no:cacheprovider -rP -k "sleap_trained or configs4_resnet50_bottomup" > $O/pytest_printed.log 2>&1; grep -a "SLEAP-trained\|configs\[4\]\|passed\|failed" $O/pytest_printed.log | cut -c1-600
for v in 1 0 1 0; do
  SA_POOL_TILED=$v timeout 300 python tools/net_profile.py resnet50 1024 16 > $O/resnet_pool$v.txt 2>&1
  echo "SA_POOL_TILED=$v"; grep -a "poolg\|^total" $O/resnet_pool$v.txt | cut -c1-160
done
