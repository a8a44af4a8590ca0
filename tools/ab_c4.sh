#!/bin/bash
# Times the diffuse-bounce workload (soup-10M, 1920x1080) with alternative builds of the library: tools/ab_c4.sh <lib>...
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root" || exit 1
for lib in "$@"; do
  echo "== $lib"
  RAYS_HIP_LIB="$root/raytracing_amd/lib/$lib" timeout -k 10 300 python3 tools/config4_probe.py 10000000 1920 1080 2>&1 | grep -v "^soup\|amdgpu.ids"
done
