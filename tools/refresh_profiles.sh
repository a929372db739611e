#!/bin/bash
# Runs ON THE GPU BOX (gpurun): bench line, rocprofv3 kernel trace of the same command, and the PMC passes
# (one rocprofv3 run per pass; counters never mixed with trace domains).  Outputs land in gpurun_out/refresh/;
# tools/fold_profiles.py (run afterwards, in the build container) folds them into profiles/.
# usage: bash tools/refresh_profiles.sh <tag>
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/refresh
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 python3 $R/bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err || exit 1
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-baseline > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu > $OUT/pmc_write.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu > $OUT/pmc_sq.log 2>&1 || exit 1
tail -c 400 $OUT/bench.json
