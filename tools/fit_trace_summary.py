#!/usr/bin/env python3
"""Histogram of the likelihood launches of a fit by size (RELATE_FIT_TRACE=1 output on stderr).
usage: tools/fit_trace_summary.py trace.log [M]"""
import re
import sys
M = float(sys.argv[2]) if len(sys.argv) > 2 else 5e5
rows = []
for l in open(sys.argv[1]):
    m = re.match(r"like launch: (\d+) lanes, (\d+) chunks, ([\d.]+) ms", l)
    if m:
        rows.append((int(m.group(1)), int(m.group(2)), float(m.group(3))))
edges = [0, 500, 2000, 5000, 10000, 20000, 45056, 65536, 98304, 131072, 196608, 262144, 10 ** 9]
for lo, hi in zip(edges, edges[1:]):
    sel = [r for r in rows if lo < r[0] <= hi]
    if sel:
        ms, pts = sum(r[2] for r in sel), sum(r[0] for r in sel)
        print("%7d..%-10d launches %4d  ms %8.1f  avg ms %6.1f  points %9d  G point-steps/s %6.1f  avg chunks %.0f" % (
            lo, hi, len(sel), ms, ms / len(sel), pts, pts * M / ms / 1e6, sum(r[1] for r in sel) / len(sel)))
print("launches %d, kernel ms %.1f" % (len(rows), sum(r[2] for r in rows)))
