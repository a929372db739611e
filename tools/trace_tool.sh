#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel trace of a python tool; summary table into gpurun_out/<tag>/kernel_stats.txt
# usage: bash tools/trace_tool.sh <tag> <script.py> [args]
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 250 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/"$@" > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
DB=$(find $OUT/trace -name '*.db' | head -1)
python3 $R/tools/rocpd_summary.py $DB > $OUT/kernel_stats.txt && rm -rf $OUT/trace
grep "relate::" $OUT/kernel_stats.txt | cut -c1-150
