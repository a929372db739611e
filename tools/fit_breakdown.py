#!/usr/bin/env python3
"""Where the time of a fitted block goes (runs ON THE GPU BOX): relate_hip_fit of one block with RELATE_FIT_TRACE=1,
every likelihood launch with its slots, the points the optimisers really asked for, its route and its HIP-event time.

The wall-clock of the call is split into
  (1) nominal FP64 issue time of the points asked for: points x M x 27.3 vector instructions / 39.3e12 per second
      (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz; profiles/r02_pmc_like_sweeps.txt for the 27.3),
  (2) the power limit: the same work at the 28.8e12 a pure v_fma_f64 stream sustains on this part
      (tools/ubench/fp64_ops.hip) minus (1),
  (3) padding: slots launched beyond the points asked for (pairs padded to the lane width, whole workgroups),
  (4) bracket under-fill and route overhead of the sweeps: a sweep costs the time of its fullest SIMD, the two-sided
      form adds a join -- the sweeps' measured time minus what their slots cost at the sustained rate,
  (5) the SNP-chunked launches (small rounds: the tail of a block, and its very first rounds): their measured time
      minus what their points cost at the sustained rate of (2) (three row vectors per lane: 53.5 instructions),
  (6) host: wall-clock minus the kernel time of all launches (optimiser steps, lane layout, transfers, restaging).
Usage: python tools/fit_breakdown.py [--pairs 16384] [--first 24576] > profiles/r03_fit_breakdown.txt"""
import argparse, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
NOMINAL, SUSTAINED, VALU1, VALU3 = 39.3e12, 28.8e12, 27.3, 53.5

if os.environ.get("RELATE_FIT_BREAKDOWN_CHILD") != "1":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=16384)
    ap.add_argument("--first", type=int, default=24576)
    ap.add_argument("--snps", type=int, default=500000)
    ap.add_argument("--inds", type=int, default=2000)
    a = ap.parse_args()
    env = dict(os.environ, RELATE_FIT_BREAKDOWN_CHILD="1", RELATE_FIT_TRACE="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--pairs", str(a.pairs), "--first", str(a.first), "--snps",
                        str(a.snps), "--inds", str(a.inds)], env=env, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stdout[-3000:] + r.stderr[-3000:])
        raise SystemExit(r.returncode)
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        open(os.path.join(ROOT, "gpurun_out", "fit_trace_raw.txt"), "w").write(r.stderr)
    head = [l for l in r.stdout.splitlines() if l.startswith("fit:")][-1]
    wall, M, P = float(re.search(r"wall ([\d.]+) s", head).group(1)), int(re.search(r"M (\d+)", head).group(1)), a.pairs
    rows, blk, mark = [], None, False
    for l in r.stderr.splitlines():
        if l.startswith("MARK"):
            mark, rows = True, []          # the launches of the warm-up fit come before the mark
        m = re.match(r"like launch: (\d+) lanes, (\d+) chunks, ([\d.]+) ms, (\d+) points of (\d+) wanted, nv (\d+), t ([\d.]+)", l)
        if m and mark:
            rows.append((int(m.group(1)), int(m.group(2)), float(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)),
                         float(m.group(7))))
        if l.startswith("fit block") and mark:
            blk = l
    print(head)
    print(blk)
    sw = [x for x in rows if x[1] == 1]
    ch = [x for x in rows if x[1] > 1]
    pts = sum(x[3] for x in rows)
    t_nom = pts * M * VALU1 / NOMINAL
    t_sus = pts * M * VALU1 / SUSTAINED
    k_sw, k_ch = sum(x[2] for x in sw) / 1e3, sum(x[2] for x in ch) / 1e3
    pad_sw = sum((x[0] - x[3]) / x[0] * x[2] for x in sw) / 1e3
    pad_ch = sum((x[0] - x[3]) / x[0] * x[2] for x in ch) / 1e3
    sus_sw = sum(x[3] for x in sw) * M * VALU1 / SUSTAINED
    sus_ch = sum(x[3] for x in ch) * M * VALU1 / SUSTAINED
    host = wall - k_sw - k_ch
    print("launches: %d sweeps (one or two passes per point) %.2f s, %d cut along the SNP axis %.2f s; points asked for %.4g "
          "(x M = %.4g point-steps), slots launched %.4g" % (len(sw), k_sw, len(ch), k_ch, pts, pts * M, sum(x[0] for x in rows)))
    parts = [("(1) nominal FP64 issue time of the points asked for", t_nom),
             ("(2) power limit (28.8 T sustained against 39.3 T nominal)", t_sus - t_nom),
             ("(3) padding slots (sweeps %.3f s, chunked %.3f s)" % (pad_sw, pad_ch), pad_sw + pad_ch),
             ("(4) sweeps: bracket under-fill, two-sided join, record traffic", k_sw - pad_sw - sus_sw),
             ("(5) SNP-chunked launches above the sustained single-vector rate", k_ch - pad_ch - sus_ch),
             ("(6) host between launches (optimiser steps, layout, transfers, restage)", host)]
    for name, v in parts:
        print("  %-78s %7.3f s  %5.1f %%" % (name, v, 100 * v / wall))
    print("  %-78s %7.3f s  (wall %.3f s = %.0f pairs/s)" % ("sum", sum(v for _, v in parts), wall, P / wall))
    # the sweeps by bracket
    edges = [0, 45056, 65536, 98304, 131072, 196608, 262144, 393216, 10 ** 9]
    print("sweeps by slot bracket (slots launched; fill = slots / bracket's upper edge; G point-steps/s over the points asked for):")
    for lo, hi in zip(edges, edges[1:]):
        sel = [x for x in sw if lo < x[0] <= hi]
        if sel:
            ms = sum(x[2] for x in sel)
            print("  %7d..%-7d launches %4d  %8.1f ms  avg %6.1f ms  mean fill %.2f  padding %.1f %%  %6.0f G/s" % (
                lo, min(hi, 999999), len(sel), ms, ms / len(sel), sum(x[0] for x in sel) / len(sel) / min(hi, 393216),
                100 * sum(x[0] - x[3] for x in sel) / sum(x[0] for x in sel), sum(x[3] for x in sel) * M / ms / 1e6))
    edges = [0, 500, 2000, 5000, 10000, 20000, 45056, 10 ** 9]
    print("launches cut along the SNP axis by size:")
    for lo, hi in zip(edges, edges[1:]):
        sel = [x for x in ch if lo < x[0] <= hi]
        if sel:
            ms = sum(x[2] for x in sel)
            print("  %7d..%-7d launches %4d  %8.1f ms  avg %6.1f ms  %6.0f G point-steps/s  (avg chunks %.0f)" % (
                lo, min(hi, 999999), len(sel), ms, ms / len(sel), sum(x[3] for x in sel) * M / ms / 1e6,
                sum(x[1] for x in sel) / len(sel)))
    # host gaps: time between the end stamp of one launch's wait and the next, minus that launch's kernel time
    gaps = [max(0.0, (b[6] - a_[6]) - b[2] / 1e3) for a_, b in zip(rows, rows[1:])]
    if gaps:
        gs = sorted(gaps)
        print("gap between consecutive launches not covered by the later one's kernel: total %.3f s, median %.1f ms, 90 %% %.1f ms, "
              "max %.1f ms over %d launches" % (sum(gaps), 1e3 * gs[len(gs) // 2], 1e3 * gs[int(0.9 * len(gs))], 1e3 * gs[-1], len(gaps)))
    raise SystemExit(0)

import numpy as np
import bench, relate_amd
from relate_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int); ap.add_argument("--first", type=int); ap.add_argument("--snps", type=int); ap.add_argument("--inds", type=int)
a = ap.parse_args()
opt = bench.opts_all_pair(); opt["back2"] = 2 * opt["back"]
geno, pos, chr_, _ = synth.make_dataset(a.inds, a.snps, device="cuda:0")
H = relate_amd.Relate(0)
M = H.init_globals(geno, pos, chr_, opt)
del geno
allp = synth.all_pairs(a.inds)
# a first, smaller block on the same handle: grows the buffers and lets the handle learn that first waves of restarts do
# not settle pairs in this mode (what every block of a scan but the first sees)
H.fit(allp[:4096])
sys.stderr.write("MARK\n"); sys.stderr.flush()
blk = allp[a.first:a.first + a.pairs]
t0 = time.perf_counter()
res = H.fit(blk)
dt = time.perf_counter() - t0
print("fit: relate_hip_fit of %d pairs (a scan's later block: restarts ahead of need from the first round), M %d, wall %.3f s, "
      "%.0f pairs/s, restarts per pair %.2f" % (a.pairs, M, dt, a.pairs / dt, sum(r["timesRun"] for r in res) / a.pairs))
