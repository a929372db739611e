#!/usr/bin/env python3
"""GPU box: the two-sided likelihood sweep against the single sweep on the same lanes (one process each, the switch is
read once): values and milliseconds.  usage: two_sided_check.py <out.npy> [pairs] [points] [snps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, relate_amd
from relate_amd import synth
out = sys.argv[1]
P = int(sys.argv[2]) if len(sys.argv) > 2 else 13000
E = int(sys.argv[3]) if len(sys.argv) > 3 else 5
M = int(sys.argv[4]) if len(sys.argv) > 4 else 100003
opt = bench.opts_all_pair(); opt["back2"] = 2 * opt["back"]
geno, pos, chr_, _ = synth.make_dataset(400, M, device="cuda:0")
H = relate_amd.Relate(0)
m = H.init_globals(geno, pos, chr_, opt)
blk = synth.all_pairs(400)[:P]
par = bench.eval_points(P, E)
par[::7, 1, 1] = 0.9; par[::7, 1, 2] = 0.9          # rejected points (k0 < 0)
par[1::7, 2, 1] = 0.0                               # k2 = 0
par[2::7, 3, 2] = 0.0                               # k1 = 0
par[3::7, 4, 1] = 0.0; par[3::7, 4, 2] = 1.0        # k = (0, 1, 0)
par[4::7, 0, 1] = 0.0; par[4::7, 0, 2] = 0.0        # k = (1, 0, 0)
v = H.like(blk, par)
ms = H.bench_like(blk, par, 3)
np.save(out, np.asarray(v))
bad = ~np.isfinite(np.asarray(v)); print("non-finite:", bad.sum(), "by point", bad.sum(axis=0), "rows", np.where(bad.any(axis=1))[0][:10] % 7)
print("two_sided=%s lanes=%d M=%d chunks=%s  %.3f ms per launch" % (os.environ.get("RELATE_LIKE_TWO_SIDED", "1"), P * E, m,
      H.last_chunks(), ms), flush=True)
