# usage: tools/isa_kernel.sh <out-prefix> <mangled-name-regex> [extra hipcc flags...]: compile csrc/conv3x3.hip to ISA, cut one kernel out
out=$1; pat=$2; shift 2
cd /root/repo/sleap_amd/csrc && /opt/rocm/bin/hipcc -S --cuda-device-only -O3 -std=c++17 --offload-arch=gfx950 -fno-honor-nans -DSA_HALF_FP16=1 "$@" -I../../include conv3x3.hip -o $out.all.s 2>&1 | grep -v "hip-link"
awk -v pat="^$pat:" '$0 ~ pat {f=1} f{print} /s_endpgm/{if(f){exit}}' $out.all.s > $out.s
grep -n "; NumVgprs\|; Occupancy\|ScratchSize\|; NumSgprs" $out.s
