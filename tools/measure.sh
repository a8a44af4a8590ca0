#!/bin/bash
# Collects the measurement artefacts of one build on a GPU box: default bench line, rocprofv3 kernel
# statistics, PMC passes (one per counter block, each a run of its own).
# usage: tools/measure.sh <tag> [extra bench.py arguments for the profiled runs]
root="$(cd "$(dirname "$0")/.." && pwd)"
tag=${1:-cur}; shift
out="$root/gpurun_out/$tag"
mkdir -p "$out"
cd "$root" || exit 1
timeout -k 10 400 python3 -u "$root/bench.py" > "$out/bench.json" 2> "$out/bench.err" || exit 1
tail -1 "$out/bench.json"
(cd /tmp && export TMPDIR=/tmp && timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extras "$@" > "$out/stats.log" 2>&1) || exit 2
"$root/tools/pmc.sh" "$tag/pmc_fetch" FETCH_SIZE -- "$@" || exit 3
"$root/tools/pmc.sh" "$tag/pmc_write" WRITE_SIZE -- "$@" || exit 4
"$root/tools/pmc.sh" "$tag/pmc_tcc" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- "$@" || exit 5
"$root/tools/pmc.sh" "$tag/pmc_sq1" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS -- "$@" || exit 6
"$root/tools/pmc.sh" "$tag/pmc_sq2" SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU -- "$@" || exit 7
python3 "$root/tools/pmc_read.py" $(find "$out" -name '*counter_collection.csv' | sort) > "$out/pmc_summary.txt"
python3 "$root/tools/stats_read.py" "$out/stats" > "$out/kernel_stats.txt"
cat "$out/kernel_stats.txt"
