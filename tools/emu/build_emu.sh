#!/bin/bash
# TEST INFRASTRUCTURE ONLY: builds the CPU-emulated twin of libmi355unet3d.so (same C ABI, kernels run by tools/emu/emu.h).
set -e
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
src="$root/3dunetcnn_amd/csrc"
objs=()
for f in "$src"/*.hip; do
  o="$here/build/$(basename "$f" .hip).o"
  mkdir -p "$here/build"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$here/emu.h" -nt "$o" ] || [ "$src/hipcompat.h" -nt "$o" ] || [ "$root/include/mi355_unet3d.h" -nt "$o" ]; then
    g++ -std=c++20 -O2 -g -fPIC -DMI355_EMU -Wno-unknown-pragmas -I"$here" -I"$src" -x c++ -c "$f" -o "$o" &
  fi
  objs+=("$o")
done
wait
g++ -std=c++20 -O2 -g -fPIC -c "$here/emu.cpp" -o "$here/build/emu.o"
g++ -shared -o "$here/libmi355unet3d_emu.so" "${objs[@]}" "$here/build/emu.o" -lpthread
echo "built $here/libmi355unet3d_emu.so"
