#!/bin/bash
# Runs ON THE GPU BOX: one rocprofv3 PMC pass (SQ instruction mix) over one bench step; per-kernel table to gpurun_out/<tag>/.
# usage: bash tools/pmc_sq.sh <tag> [bench args]
set -o pipefail
TAG=${1:-pmc}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu "$@" > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
CSV=$(find $OUT/pmc -name '*counter_collection.csv' | head -1)
python3 - "$CSV" > $OUT/pmc_sq.txt <<'PY'
import csv, sys, collections
rows = collections.OrderedDict()
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        if "relate::" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        rows.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])   # last dispatch wins
names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
print("%-50s " % "kernel" + " ".join("%16s" % n[-16:] for n in names))
for k, v in rows.items():
    print("%-50s " % k[-50:] + " ".join("%16.4g" % v.get(n, float("nan")) for n in names))
PY
rm -rf $OUT/pmc
cat $OUT/pmc_sq.txt | cut -c1-220
