# A/B builds: libsleap_amd_fp16 with ONE translation unit recompiled under extra defines.
#   tools/build_alt.sh <name> <file.hip> "<-D...>"   -> sleap_amd/lib/alt/libsleap_amd_fp16_<name>.so  (use: SLEAP_AMD_LIB_FP16=<path>)
set -e
S=/root/repo/sleap_amd
mkdir -p $S/lib/alt
extra=""; [ "$2" = conv3x3.hip ] && extra="-fno-honor-nans"; [ "$2" = stem16.hip ] && extra="-fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc -c $S/csrc/$2 -o $S/lib/alt/$1.o -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $extra -DSA_HALF_FP16=1 $3 -I/root/repo/include
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $S/lib/alt/libsleap_amd_fp16_$1.so $S/lib/alt/$1.o $(ls $S/lib/fp16/*.o | grep -v "/${2%.hip}.o")
ls -la $S/lib/alt/libsleap_amd_fp16_$1.so
