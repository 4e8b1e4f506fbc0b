#!/bin/bash
# Developer tool: build a VARIANT of the kernel library for same-box A/B runs (tools/bench_conv_layers.py, tools/ab_lib.py).
#   tools/build_variant.sh <tag> <source.hip> [-DNAME=value ...]     -> tools/libvar_<tag>.so
#   tools/build_variant.sh head                                       -> tools/libvar_head.so from the sources of git HEAD
# One .hip is recompiled with the given switches and linked with the up-to-date objects of the other sources (run
# `python 3dunetcnn_amd/build.py` first). Variant libraries are git-ignored and travel to the GPU box with the tree; delete them after.
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
tag=$1; shift
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
if [ "$tag" = head ]; then
  tmp=$(mktemp -d)
  git -C "$root" archive HEAD 3dunetcnn_amd/csrc include | tar -x -C "$tmp"
  (cd "$tmp/3dunetcnn_amd/csrc" && ls *.hip | xargs -P 9 -I{} /opt/rocm/bin/hipcc $flags -c {} -o {}.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/libvar_head.so" *.o)
  rm -rf "$tmp"
  echo "$root/tools/libvar_head.so"
  exit 0
fi
src=$1; shift
base=$(basename "$src" .hip)
obj=$(mktemp --suffix=.o)
/opt/rocm/bin/hipcc $flags "$@" -c "$root/3dunetcnn_amd/csrc/$base.hip" -o "$obj" -Rpass-analysis=kernel-resource-usage 2> "$root/tools/libvar_$tag.resources.txt"
others=$(ls "$root"/3dunetcnn_amd/csrc/build/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/libvar_$tag.so" $others "$obj"
rm -f "$obj"
echo "$root/tools/libvar_$tag.so  (register / occupancy remarks: tools/libvar_$tag.resources.txt; spills: $(grep -c 'VGPRs Spill: [1-9]' "$root/tools/libvar_$tag.resources.txt" || true))"
