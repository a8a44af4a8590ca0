#!/bin/bash
for b in 14 16 18 20 22; do
echo "== bits $b"
RAYS_HIP_SORT_BITS=$b timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['config']['phase_ms'].items()})"
done
