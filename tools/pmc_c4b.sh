#!/bin/bash
# Memory-pipeline counter passes over the diffuse-bounce workload (see tools/pmc_c4.sh).
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/c4pmc"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 "$root/tools/config4_probe.py" 10000000 1920 1080 > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$out/$name.log"; exit 2; }; }
pass ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE && pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE || exit 2
python3 "$root/tools/pmc_read.py" $(find "$out/ta" "$out/tlb" -name "*counter_collection.csv") > "$out/summary_b.txt"
grep -A16 "^k_wf_bounce<false>" "$out/summary_b.txt"
