# round 4, GPU call C: tap-GEMM tile shape / chunk size sweep on the ResNet-50 network (planes), correctness of the forced
# variants, the tiled-grayscale stem, SA_CONV_NT on the concatenated decoder convs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04c}; mkdir -p $O; cd $R
for sh in 0 1 2 3; do for ck in 64 32; do
  export SA_TAP_SHAPE=$sh SA_TAP_CK=$ck
  timeout 300 python tools/net_profile.py resnet50 1024 16 0.012 > $O/rn_s${sh}_c${ck}.txt 2>&1
  echo "shape $sh ck $ck: $(grep -E '^conv1x1s1|^conv1x1s2|^convT4|^total' $O/rn_s${sh}_c${ck}.txt | cut -c1-60 | tr '\n' '|')" | tee -a $O/sweep.txt
done; done
export SA_TAP_SHAPE=3 SA_TAP_CK=32
timeout 600 python -m pytest tests/test_gpu_backbones.py -m gpu -q -k "resnet or convt or conv1x1 or stem_block" -p no:cacheprovider 2>&1 | tail -n 5 | cut -c1-200
export SA_TAP_SHAPE=1 SA_TAP_CK=32
timeout 600 python -m pytest tests/test_gpu_backbones.py -m gpu -q -k "resnet or convt or conv1x1 or stem_block" -p no:cacheprovider 2>&1 | tail -n 5 | cut -c1-200
unset SA_TAP_SHAPE SA_TAP_CK
timeout 600 python -m pytest tests/test_gpu_backbones.py tests/test_gpu_config_parity.py -m gpu -q -p no:cacheprovider -s 2>&1 | grep -E "configs\[4\]|two-stack|passed|failed" | cut -c1-400
export SA_CONV_NT=0
timeout 300 python tools/net_profile.py resnet50 1024 16 0.012 > $O/rn_nt0.txt 2>&1; grep -E "conv3x3 .*mode1|^total" $O/rn_nt0.txt | cut -c1-110
unset SA_CONV_NT
grep -E "conv3x3 .*mode1|imgconv|^total" $O/rn_s0_c64.txt | cut -c1-110
