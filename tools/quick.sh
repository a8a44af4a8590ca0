#!/bin/bash
# Quick loop on a GPU box: the projection / frame tests, a short bench line, kernel times.
# usage: tools/quick.sh <tag> [pytest -k expression]
root="$(cd "$(dirname "$0")/.." && pwd)"
tag=${1:-q}; kexpr=${2:-"projection or slivers or config3 or soup_primary_and_shadow"}
out="$root/gpurun_out/$tag"; mkdir -p "$out"; cd "$root" || exit 1
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q -k "$kexpr" --timeout 300 --timeout-method thread > "$out/pytest.log" 2>&1
echo "pytest rc=$?"; tail -4 "$out/pytest.log"
(cd /tmp && export TMPDIR=/tmp && timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extras > "$out/bench.json" 2> "$out/bench.err")
python3 -c "
import json,sys
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['config']['phase_ms'].items()}, d['config'].get('camera_phase_counters'))"
python3 "$root/tools/stats_read.py" "$out/stats" 2>/dev/null | grep -v "Lb1E\|<true" | sed -n 1,12p
