#!/bin/bash
# usage: tools/build_ref_variant.sh <git-rev> <name>  -> q-diffusion_amd/lib/libqdiff_hip_<name>.so built from that revision's csrc
# (A/B measurements on one GPU box: QDIFF_HIP_LIB=<that file>; only meaningful while the C ABI is unchanged)
set -e
rev=$1; name=$2
root=$(git rev-parse --show-toplevel)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" q-diffusion_amd/csrc include | tar -x -C "$tmp"
mkdir -p "$tmp/obj"
for f in "$tmp"/q-diffusion_amd/csrc/*.hip "$tmp"/q-diffusion_amd/csrc/*.cpp; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-value -x hip -c "$f" -o "$tmp/obj/$(basename "$f").o" &
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o "$root/q-diffusion_amd/lib/libqdiff_hip_$name.so" "$tmp"/obj/*.o
rm -rf "$tmp"
echo "$root/q-diffusion_amd/lib/libqdiff_hip_$name.so"
