#!/bin/bash
# Diffuse-bounce frame (soup-N, 3840x2160, 10 scatter rays per hit) with alternative builds of the library.
# usage: tools/ab4.sh <triangles> <lib>...
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root" || exit 1
N=$1; shift
for lib in "$@"; do
  echo "== $lib"
  RAYS_HIP_LIB="$root/raytracing_amd/lib/$lib" RAYS_HIP_BOUNCE_RAYS=1400000000 timeout -k 10 400 python3 tools/config4_probe.py $N 3840 2160 2>&1 | grep -v "amdgpu.ids\|projection would\|camera kernel" || exit 1
done
