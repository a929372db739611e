#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel trace of the bench step (no CPU legs), summary table into gpurun_out/<tag>/.
# usage: bash tools/trace_bench.sh <tag> [bench args]
set -o pipefail
TAG=${1:-trace}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu "$@" > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
DB=$(find $OUT/trace -name '*.db' | head -1)
python3 $R/tools/rocpd_summary.py $DB > $OUT/kernel_stats.txt && rm -rf $OUT/trace
head -16 $OUT/kernel_stats.txt | cut -c1-150
