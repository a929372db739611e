#!/bin/bash
# Runs ON THE GPU BOX: one rocprofv3 PMC pass over a python tool; per-kernel table (last dispatch of each relate:: kernel).
# usage: bash tools/pmc_run.sh <tag> "<counters>" <script.py> [args]      (the program after -- is python3 itself)
set -o pipefail
TAG=$1; CTR=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc $CTR --output-format csv -d $OUT/pmc -o c -- python3 $R/"$@" > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
CSV=$(find $OUT/pmc -name '*counter_collection.csv' | head -1)
python3 - "$CSV" "$CTR" > $OUT/pmc_table.txt <<'PY'
import csv, sys, collections
rows = collections.OrderedDict()
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        if "relate::" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("relate::", "")
        rows.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])   # last dispatch wins
names = sys.argv[2].split()
print("%-40s " % "kernel" + " ".join("%16s" % n[-16:] for n in names))
for k, v in rows.items():
    print("%-40s " % k[-40:] + " ".join("%16.5g" % v.get(n, float("nan")) for n in names))
PY
rm -rf $OUT/pmc
grep -v "pack_\|maf_\|gather_\|table_\|ld_\|choose_\|transpose" $OUT/pmc_table.txt | cut -c1-200
