#!/bin/bash
# Runs ON THE GPU BOX: tools/fitbench.py under several settings of the fit's scheduling switches, one line + launch
# histogram each.  usage: bash tools/fit_variants.sh <tag> <pairs> "<eager> <pops>" ...   (eager/pops: 0|1|2|x = unset)
set -o pipefail
TAG=$1; PAIRS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for v in "$@"; do
  set -- $v
  unset RELATE_FIT_EAGER RELATE_FIT_POPULATIONS
  [ "$1" != x ] && export RELATE_FIT_EAGER=$1
  [ "$2" != x ] && export RELATE_FIT_POPULATIONS=$2
  echo "== eager=$1 pops=$2"
  RELATE_FIT_TRACE=1 timeout -k 10 200 python3 $R/tools/fitbench.py --inds 2000 --snps 500000 --pairs $PAIRS \
    > $OUT/fit_$1_$2.log 2> $OUT/trace_$1_$2.log || { tail -3 $OUT/trace_$1_$2.log; exit 1; }
  cat $OUT/fit_$1_$2.log
  python3 $R/tools/fit_trace_summary.py $OUT/trace_$1_$2.log | tail -7
done
