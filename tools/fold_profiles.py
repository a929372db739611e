#!/usr/bin/env python3
"""Folds gpurun_out/refresh/ (tools/refresh_profiles.sh) into profiles/: <tag>_bench.json, <tag>_bench_kernel_stats.txt,
pmc_traffic.json (FETCH_SIZE / WRITE_SIZE passes) and <tag>_pmc_sq.json (SQ pass: per kernel VALU instructions per
wave-step inputs, VALU utilisation and the clock the part held = GRBM_GUI_ACTIVE / 8 / duration).
usage: tools/fold_profiles.py <tag>"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")
DST = os.path.join(ROOT, "profiles")


def last_dispatch(path, names):
    """Last dispatch of every (kernel, grid size): the likelihood and the decode launch of chunk_matrix_kernel differ
    in their grids and are kept apart."""
    per = {}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if "relate::" not in r["Kernel_Name"] or r["Counter_Name"] not in names:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            per.setdefault((int(r["Dispatch_Id"]), k, r.get("Grid_Size", "")), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    out = {}
    for (disp, k, grid), v in sorted(per.items()):
        out["%s grid=%s" % (k, grid)] = v  # later dispatches overwrite
    return out


def main(tag):
    line = open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1]
    bench = json.loads(line)
    with open(os.path.join(DST, tag + "_bench.json"), "w") as fh:
        fh.write(line + "\n")
    db = glob.glob(os.path.join(SRC, "trace", "**", "*results.db"), recursive=True)
    if db:
        txt = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), db[0]], text=True)
        with open(os.path.join(DST, tag + "_bench_kernel_stats.txt"), "w") as fh:
            fh.write(txt)
    cfg = bench["config"]["workload"]
    P = int(cfg.split(" pairs/GPU/step")[0].split()[-1])
    M = int(cfg.split(" after MAF filter")[0].split("(")[-1])
    E = int(cfg.split("E=")[1].split()[0])
    f = glob.glob(os.path.join(SRC, "pmc_fetch", "**", "*counter_collection.csv"), recursive=True)
    w = glob.glob(os.path.join(SRC, "pmc_write", "**", "*counter_collection.csv"), recursive=True)
    if f and w:
        txt = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), f[0], w[0],
                                       str(P), str(M), str(E)], text=True)
        with open(os.path.join(DST, "pmc_traffic.json"), "w") as fh:
            fh.write(txt)
    q = glob.glob(os.path.join(SRC, "pmc_sq", "**", "*counter_collection.csv"), recursive=True)
    if q:
        names = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                 "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE"]
        d = last_dispatch(q[0], names)
        res = {"_comment": "rocprofv3 --pmc (one pass) of `bench.py --steps 1 --warmup 1 --no-cpu`; last dispatch of each "
                           "kernel.  SQ_* cycle counters are in quad-cycles summed over all waves; cycles = "
                           "GRBM_GUI_ACTIVE / 8 (sum over the 8 XCDs); valu_util = SQ_ACTIVE_INST_VALU / (1024 SIMDs x "
                           "cycles / 4); wait_any / wait_inst = share of wave residency parked on s_waitcnt / stalled at "
                           "issue.", "P": P, "M": M, "E": E, "kernels": {}}
        for k, v in d.items():
            cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
            wc = v.get("SQ_WAVE_CYCLES", 0) or 1
            res["kernels"][k] = dict(v, cycles=cyc,
                                     valu_util=(v.get("SQ_ACTIVE_INST_VALU", 0) / (1024 * cyc / 4)) if cyc else None,
                                     wait_any=v.get("SQ_WAIT_ANY", 0) / wc, wait_inst=v.get("SQ_WAIT_INST_ANY", 0) / wc)
        with open(os.path.join(DST, tag + "_pmc_sq.json"), "w") as fh:
            json.dump(res, fh, indent=1)
    print("folded into", DST)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
