#!/bin/bash
# One rocprofv3 counter pass over a short bench run (run on the GPU box).
# usage: tools/pmc.sh <outdir-name under gpurun_out> <counters...> [-- extra bench.py arguments]
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; shift
counters=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do counters+=("$1"); shift; done
[ "$1" = "--" ] && shift
mkdir -p "$root/gpurun_out/$(dirname "$name")"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "${counters[@]}" --output-format csv -d "$root/gpurun_out/$name" -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > "$root/gpurun_out/$name.log" 2>&1
