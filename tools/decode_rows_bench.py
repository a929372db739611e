#!/usr/bin/env python3
"""K6 timing by posterior rows written (runs ON THE GPU BOX): the bench's pair block, decode with the reference's three
rows (rows = 0) and with 1, 2, 3 rows in the writer's order; HIP-event ms of pass A and pass B, 3 launches each."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, relate_amd
from relate_amd import synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
opt = bench.opts_all_pair(); opt["back2"] = 2 * opt["back"]
geno, pos, chr_, _ = synth.make_dataset(2000, 500000, device="cuda:0")
H = relate_amd.Relate(0)
M = H.init_globals(geno, pos, chr_, opt)
del geno
blk = synth.all_pairs(2000)[:P]
par = np.ascontiguousarray(bench.eval_points(P, 1)[:, 0, :])
ROWS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (0, 3, 2, 1, 0, 2)
for rows in ROWS:
    H.bench_decode(blk, par, 1, rows=rows)
    f, b = H.bench_decode(blk, par, 3, rows=rows)
    nb = 3 if rows == 0 else rows
    print("rows %d (%s), %d chunks: pass A %.2f ms, pass B %.2f ms = %.2f TB/s on %.2f B per pair and SNP" % (
        rows, "reference order" if rows == 0 else "writer order", H.last_chunks()[1], f, b, (16 + 5.25 + 8 * nb) * P * M / b / 1e9, 16 + 5.25 + 8 * nb), flush=True)
