#!/bin/bash
# cells against tree-walk shadow rays at several frame sizes and triangle counts
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root" || exit 1
for cfg in "1000000 1920 1080" "1000000 960 540" "300000 3840 2160" "3000000 3840 2160" "200000 1280 720"; do
  set -- $cfg
  for mode in cells-default walk; do
    if [ $mode = walk ]; then export RAYS_HIP_SHADOW=walk; else unset RAYS_HIP_SHADOW; fi
    timeout -k 10 200 python3 bench.py --tris $1 --width $2 --height $3 --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 tris $2x$3 $mode:', round(d['ms_per_step'],3), 'ms', d['config']['shadow_rays_by'], d['config']['camera_rays_by'][:10], {k: round(v,3) for k,v in d['config']['phase_ms'].items()})" || exit 1
  done
done
