#!/bin/bash
# Collects the measurement artefacts of one build on a GPU box (run from the repo root):
# default bench line, rocprofv3 kernel statistics, PMC passes (one per counter block), parity report.
# usage: tools/measure.sh <tag>
tag=${1:-cur}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
timeout -k 10 300 python3 -u bench.py > $out/bench.json 2> $out/bench.err || exit 1
tail -1 $out/bench.json
(cd /tmp && export TMPDIR=/tmp && timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/stats.log 2>&1) || exit 2
tools/pmc.sh $tag/pmc_fetch FETCH_SIZE || exit 3
tools/pmc.sh $tag/pmc_write WRITE_SIZE || exit 4
tools/pmc.sh $tag/pmc_tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 5
tools/pmc.sh $tag/pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE || exit 6
tools/pmc.sh $tag/pmc_sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU || exit 7
python3 tools/pmc_read.py $(find $out -name '*counter_collection.csv' | sort) > $out/pmc_summary.txt
timeout -k 10 400 python3 -u tools/parity_report.py > $out/parity.json 2> $out/parity.err || exit 8
tail -3 $out/parity.json
