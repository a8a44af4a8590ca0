#!/bin/bash
# Kernel times of the bench frame with an alternative build of the library: tools/abprof.sh <lib> [filter]
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root" || exit 1
out="$root/gpurun_out/abprof_$1"; mkdir -p "$out"
(cd /tmp && export TMPDIR=/tmp && RAYS_HIP_LIB="$root/raytracing_amd/lib/$1" timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$out/bench.json" 2> "$out/bench.err")
python3 "$root/tools/stats_read.py" "$out/stats" 2>/dev/null | grep -i "${2:-raster}"
