#!/bin/bash
# Counter passes over the diffuse-bounce workload (tools/config4_probe.py, soup-10M, one run of tiles per frame;
# frame size from the arguments, default 3840 2160): run on the GPU box.
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/c4pmc"; mkdir -p "$out"
W=${1:-3840}; H=${2:-2160}
export RAYS_HIP_BOUNCE_RAYS=1400000000
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 "$root/tools/config4_probe.py" 10000000 $W $H > "$out/$name.log" 2>&1 || { echo "pass $name failed"; exit 2; }; }
pass fetch FETCH_SIZE TCC_HIT_sum && pass write WRITE_SIZE TCC_MISS_sum TCC_REQ_sum && pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM && pass sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES || exit 2
python3 "$root/tools/pmc_read.py" $(find "$out" -name "*counter_collection.csv") > "$out/summary.txt"
grep -A22 "^k_wf_bounce<false>" "$out/summary.txt"
