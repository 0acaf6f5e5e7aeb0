# round 4, GPU call F: tap GEMM register budget / residual prefetch A/B on the ResNet network (alternative libraries)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; L=$R/sleap_amd/lib; O=$R/gpurun_out/${1:-r04f}; mkdir -p $O; cd $R
for i in 1 2; do for v in base alt_tap_nopf alt_tap_nopf_w4 alt_tap_w4; do
  if [ $v = base ]; then unset SLEAP_AMD_LIB_FP16; else export SLEAP_AMD_LIB_FP16=$L/$v.so; fi
  timeout 300 python tools/net_profile.py resnet50 1024 16 0.012 > $O/rn_${v}_$i.txt 2>&1
  echo "$v: $(grep -E '^conv1x1s1|^conv1x1s2|^convT4|^total' $O/rn_${v}_$i.txt | cut -c1-62 | tr '\n' '|')" | tee -a $O/sweep.txt
done; done
unset SLEAP_AMD_LIB_FP16
