VF=$GRAFT_REPO_ROOT/sleap_amd/lib/libsleap_amd_fp16_vf.so
for net in resnet50 hourglass; do
for v in base vf base vf; do
  if [ $v = vf ]; then export SLEAP_AMD_LIB_FP16=$VF; else unset SLEAP_AMD_LIB_FP16; fi
  echo "== $net $v"; SLEAP_AMD_DTYPE=fp16 timeout 200 python tools/net_profile.py $net 512 16 1.0 2>&1 | tail -9
done; done
