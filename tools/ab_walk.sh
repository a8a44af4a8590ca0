#!/bin/bash
# The benchmark frame with every ray by tree walk, with alternative builds of the library: tools/ab_walk.sh <lib>...
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root" || exit 1
for lib in "$@"; do
  RAYS_HIP_LIB="$root/raytracing_amd/lib/$lib" RAYS_HIP_RASTER=0 RAYS_HIP_SHADOW=walk timeout -k 10 200 python3 tools/frame_probe.py "$lib" 2>/dev/null | cut -c1-100 || exit 1
done
