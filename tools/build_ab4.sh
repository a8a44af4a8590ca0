#!/bin/bash
# Diffuse-bounce frame (per-lane walk of the 4-wide tree) and the camera + shadow walks under the host's and
# the device's tree builders. usage: build_ab4.sh [triangles] [builders...]
set -e
N=${1:-10000000}; shift || true
for b in ${@:-host device}; do
  echo "== $b"
  RAYS_HIP_BUILD=$b RAYS_HIP_VERBOSE=1 RAYS_HIP_BOUNCE_RAYS=1400000000 timeout -k 10 400 python3 tools/config4_probe.py $N 3840 2160 2>&1 | grep -v "amdgpu.ids\|projection would\|camera kernel"
done
