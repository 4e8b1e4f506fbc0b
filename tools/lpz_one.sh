#!/bin/bash
# Developer tool: compile ONE instantiation of a plane-ring kernel and print its register / spill remarks (the whole file takes minutes).
#   tools/lpz_one.sh "conv3d_k3_lp_zring2<2, 2, 0, 1, false>" [extra hipcc flags]
root="$(cd "$(dirname "$0")/.." && pwd)"
inst=$1; shift
tmp=$(mktemp -d)
cat > $tmp/one.hip <<EOT
#define LPZ_NO_LAUNCH
#include "$root/3dunetcnn_amd/csrc/conv3d_bf16_zring.hip"
template __global__ void $inst(ConvBArgs);
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$root/3dunetcnn_amd/csrc "$@" -c $tmp/one.hip -o $tmp/one.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep "VGPRs:\|AGPRs\|Spill\|Scratch\|error" | sed 's/.*remark: //;s/\[-Rpass.*//' | paste -s
[ -n "$KEEP" ] && cp $tmp/one.o $KEEP
rm -rf $tmp
