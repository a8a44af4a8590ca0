#!/bin/bash
# Times the bench frame under settings of one environment variable: tools/ab_env.sh NAME value...
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root" || exit 1
name=$1; shift
for v in "$@"; do
  echo "== $name=$v"
  env "$name=$v" timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['config']['phase_ms'].items()})"
done
