# round 4, GPU call J: 1 x 1 convs with several cout tiles -- one workgroup per pixel tile walks the cout tiles (SA_TAP_COLOOP 1 / 0)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04j}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_backbones.py tests/test_gpu_layer_pins.py tests/test_gpu_network.py -m gpu -q --maxfail=10 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 8 $O/pytest.log | cut -c1-300
for i in 1 2; do for v in 1 0 2; do
  echo "== SA_TAP_COLOOP=$v" >> $O/resnet.txt
  SA_TAP_COLOOP=$v timeout 300 python tools/net_profile.py resnet50 1024 16 0.012 2>/dev/null >> $O/resnet.txt
done; done
grep -E "^==|^total|conv1x1s1 +n|conv1x1s2 +n|64->256 @256 \+affine|128->512 @128|256->1024 @64" $O/resnet.txt | cut -c1-140
