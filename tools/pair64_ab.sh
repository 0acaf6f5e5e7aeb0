# A/B of builds of csrc/convpair64.hip on ONE box: bash tools/pair64_ab.sh <lib1.so> ... (files in sleap_amd/lib, tools/build_alt.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; L=$R/sleap_amd/lib; cd $R
for i in 1 2 3; do
for v in base "$@"; do
  if [ $v = base ]; then unset SLEAP_AMD_LIB_FP16; else export SLEAP_AMD_LIB_FP16=$L/$v; fi
  echo "$v: $(python tools/pair64_probe.py 64 2>/dev/null | head -1) | bench layer: $(python bench.py --no-cpu-baseline --no-extras --layers --steps 20 2>&1 >/dev/null | grep '32->64->64' | cut -c45-80)"
done; done
