#!/bin/bash
# TEST INFRASTRUCTURE ONLY: builds the CPU-emulated twin of libmi355unet3d.so (same C ABI, kernels run by tools/emu/emu.h).
set -e
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
src="$root/3dunetcnn_amd/csrc"
# one builder at a time: parallel test workers all call this script; the first rebuilds, the others wait and find everything fresh
mkdir -p "$here/build"
exec 9> "$here/build/.lock"
flock 9
objs=()
pids=()
rebuilt=0
for f in "$src"/*.hip; do
  o="$here/build/$(basename "$f" .hip).o"
  mkdir -p "$here/build"
  stale=0
  # dependencies: the headers the object was compiled from (g++ -MD depfile beside it); without one, every header a source may include
  if [ -f "$o.d" ]; then deps=$(sed -e 's/^[^:]*://' -e 's/\\$//' "$o.d" | tr ' ' '\n' | grep -E "^($root|\.\./|[^/])" | sort -u)
  else deps="$f $here/emu.h $root/include/mi355_unet3d.h $(ls "$src"/*.h)"; fi
  for dep in $deps; do
    if [ ! -f "$o" ] || { [ -f "$dep" ] && [ "$dep" -nt "$o" ]; }; then stale=1; fi
  done
  if [ "$stale" = 1 ]; then
    g++ -std=c++20 -O2 -g -fPIC -DMI355_EMU -Wno-unknown-pragmas -I"$here" -I"$src" -MD -MF "$o.d" -x c++ -c "$f" -o "$o" &
    pids+=($!)
    rebuilt=1
  fi
  objs+=("$o")
done
for p in "${pids[@]}"; do wait "$p"; done      # a failed compile fails the script (plain `wait` would hide it)
so="$here/libmi355unet3d_emu.so"
if [ ! -f "$here/build/emu.o" ] || [ "$here/emu.cpp" -nt "$here/build/emu.o" ] || [ "$here/emu.h" -nt "$here/build/emu.o" ]; then
  g++ -std=c++20 -O2 -g -fPIC -c "$here/emu.cpp" -o "$here/build/emu.o"
  rebuilt=1
fi
if [ "$rebuilt" = 1 ] || [ ! -f "$so" ]; then
  # link beside the target and rename: parallel test workers that call this script never see a half-written library
  g++ -shared -o "$so.$$" "${objs[@]}" "$here/build/emu.o" -lpthread
  mv -f "$so.$$" "$so"
fi
echo "built $so"
