#!/bin/bash
# Kernel statistics of the diffuse-bounce workload at full size (soup-10M, 3840x2160, 4 samples, 10 scatter
# rays per hit): run on the GPU box; writes gpurun_out/c4/probe.txt and gpurun_out/c4/kernel_stats.txt.
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/c4"; mkdir -p "$out"
cd "$root" || exit 1
RAYS_HIP_BOUNCE_RAYS=1400000000 timeout -k 10 400 python3 tools/config4_probe.py 10000000 3840 2160 2>/dev/null > "$out/probe.txt" || exit 1
cat "$out/probe.txt"
(cd /tmp && export TMPDIR=/tmp && RAYS_HIP_BOUNCE_RAYS=1400000000 timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/tools/config4_probe.py" 10000000 3840 2160 > "$out/stats.log" 2>&1) || exit 2
python3 "$root/tools/stats_read.py" "$out/stats" > "$out/kernel_stats.txt"
sed -n 1,16p "$out/kernel_stats.txt"
