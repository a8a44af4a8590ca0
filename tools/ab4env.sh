#!/bin/bash
# tools/config4_probe.py on soup-N at 3840x2160 under environment settings given as arguments ("A=1 B=2" each).
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root" || exit 1
N=$1; shift
for e in "$@"; do
  echo "== $e"
  env $e RAYS_HIP_BOUNCE_RAYS=1400000000 timeout -k 10 400 python3 tools/config4_probe.py $N 3840 2160 2>&1 | grep "diffuse" || exit 1
done
