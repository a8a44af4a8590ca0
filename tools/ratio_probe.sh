#!/bin/bash
# camera rays by projection against the tree walk around the work limit of the projection
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root" || exit 1
for cfg in "1000000 960 540" "3000000 3840 2160" "2000000 1920 1080" "5000000 3840 2160" "1000000 640 360"; do
  set -- $cfg
  for ratio in 1e9 0; do
    RAYS_HIP_VERBOSE=1 RAYS_HIP_RASTER_RATIO=$ratio timeout -k 10 200 python3 bench.py --tris $1 --width $2 --height $3 --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2> /tmp/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 tris $2x$3 limit $ratio:', round(d['ms_per_step'],3), 'ms', d['config']['camera_rays_by'][:10], d['config']['shadow_rays_by'], {k: round(v,3) for k,v in d['config']['phase_ms'].items()})" || exit 1
    grep -m1 "projection would test" /tmp/err.txt
  done
done
