#!/bin/bash
# usage: tools/pmc.sh <outdir-name> <counters...>   (run on the GPU box from the repo root)
set -e
name=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$name -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/$name.log 2>&1
