#!/usr/bin/env python3
"""Ad-hoc kernel timing on the GPU box: like / decode / stage at several (pairs, points) shapes.
usage: python tools/kbench.py [--snps M] [--inds N] shape ...   with shape = P:E"""
import argparse
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import relate_amd  # noqa: E402
from relate_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--snps", type=int, default=100000)
ap.add_argument("--inds", type=int, default=400)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--decode", action="store_true")
ap.add_argument("--miss", type=float, default=0.001, help="missing-genotype rate of the synthetic data")
ap.add_argument("--lib", default=None, help="experiment: load this build of the library instead")
ap.add_argument("shapes", nargs="+")
a = ap.parse_args()
if a.lib:
    import relate_amd.api
    relate_amd.api.LIB_PATH = os.path.abspath(a.lib)
opt = bench.opts_all_pair()
opt["back2"] = 2 * opt["back"]
geno, pos, chr_, _ = synth.make_dataset(a.inds, a.snps, device="cuda:0", miss=a.miss)
H = relate_amd.Relate(0)
M = H.init_globals(geno, pos, chr_, opt)
allp = synth.all_pairs(a.inds)
for sh in a.shapes:
    P, E = (int(v) for v in sh.split(":"))
    blk = allp[:P]
    par = bench.eval_points(P, E)
    H.bench_like(blk, par, 1)
    st = H.timings()["stage_ms"]
    ms = H.bench_like(blk, par, a.reps)
    line = "P=%d E=%d M=%d lanes=%d  stage %.2f ms (%.0f GB/s)  like %.3f ms  %.1f G lane-steps/s  rec-read %.0f GB/s" % (
        P, E, M, P * E, st, 16.75 * P * M / st / 1e6 if st else 0, ms, P * E * M / ms / 1e6, 16.0 * P * M / ms / 1e6)
    if a.decode:
        f, b = H.bench_decode(blk, np.ascontiguousarray(par[:, 0, :]), a.reps)
        line += "  dec_fwd %.2f ms (%.0f GB/s) dec_bwd %.2f ms (%.0f GB/s)" % (
            f, 21.25 * P * M / f / 1e6, b, 45.25 * P * M / b / 1e6)
    print(line, flush=True)
