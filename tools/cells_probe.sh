#!/bin/bash
# shadow rays by light-space cells (forced) against the tree walk as the lists grow longer
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root" || exit 1
for cfg in "1500000 3840 2160" "2000000 3840 2160" "3000000 3840 2160" "2000000 1920 1080" "500000 1920 1080"; do
  set -- $cfg
  for mode in cells walk; do
    RAYS_HIP_SHADOW=$mode timeout -k 10 200 python3 bench.py --tris $1 --width $2 --height $3 --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['config']
print('$1 tris $2x$3 $mode:', round(d['ms_per_step'],3), 'ms', c['shadow_rays_by'], 'mean list', round(c['shadow_lists']['entries']/max(1,c['shadow_lists']['cells_with_a_list']),1), {k: round(v,3) for k,v in c['phase_ms'].items()})" || exit 1
  done
done
