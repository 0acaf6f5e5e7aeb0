cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03p}; mkdir -p $O; cd $R
for i in 1 2; do for v in base tap_nopf tap_lds; do
  if [ $v = base ]; then unset SLEAP_AMD_LIB_FP16; else export SLEAP_AMD_LIB_FP16=$R/sleap_amd/lib/libalt_$v.so; fi
  timeout 200 python tools/net_profile.py resnet50 1024 16 0.03 2>/dev/null | grep -E "^conv1x1s|^convT|^total" | sed "s/^/$v /" | tee -a $O/ab.txt
done; done
unset SLEAP_AMD_LIB_FP16; timeout 300 python -m pytest tests/test_gpu_backbones.py tests/test_gpu_layer_pins.py -m gpu -q -p no:cacheprovider 2>&1 | tail -n 3
