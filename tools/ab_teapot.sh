#!/bin/bash
# The default scene's frames (tools/teapot_probe.py) with alternative builds of the library: tools/ab_teapot.sh <lib>...
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root" || exit 1
for lib in "$@"; do
  echo "== $lib"
  RAYS_HIP_LIB="$root/raytracing_amd/lib/$lib" timeout -k 10 200 python3 tools/teapot_probe.py 2>/dev/null | grep "default\|reference" || exit 1
done
