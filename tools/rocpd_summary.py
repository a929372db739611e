#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) as the kernel-stats table that goes
under profiles/: name, calls, total/average/min/max duration, VGPRs, scratch.
usage: tools/rocpd_summary.py results.db > profiles/<name>.txt"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(accum_vgpr_count), max(sgpr_count), max(scratch_size), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)")
    print("%-64s %7s %12s %11s %11s %11s %6s %5s %5s %5s %7s %6s %10s %4s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "sgpr", "scratch",
        "lds", "grid_x", "wg"))
    for r in rows:
        name = r[0].split("(")[0][-64:]
        print("%-64s %7d %12.1f %11.1f %11.1f %11.1f %6.2f %5d %5d %5d %7d %6d %10d %4d" % (
            name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7], r[8], r[9],
            r[10], r[11], r[12]))


if __name__ == "__main__":
    main(sys.argv[1])
