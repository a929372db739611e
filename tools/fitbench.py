#!/usr/bin/env python3
"""End-to-end fit timing on the GPU box: relate_hip_fit on P pairs of a synthetic data set."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, relate_amd
from relate_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--snps", type=int, default=100000)
ap.add_argument("--inds", type=int, default=400)
ap.add_argument("--pairs", type=int, default=2048)
ap.add_argument("--mode", default="all")
a = ap.parse_args()
opt = bench.opts_all_pair(); opt["back2"] = 2 * opt["back"]
if a.mode == "full":
    opt.update(calcA=0, alim1=float(np.float32(0.15)))
elif a.mode == "link":
    opt.update(calcA=1, fixK=1, fixK_val=0.0, timesToRun=1, timesToConverge=1)
geno, pos, chr_, _ = synth.make_dataset(a.inds, a.snps, device="cuda:0")
H = relate_amd.Relate(0)
M = H.init_globals(geno, pos, chr_, opt)
blk = synth.all_pairs(a.inds)[:a.pairs]
t0 = time.perf_counter()
res = H.fit(blk)
dt = time.perf_counter() - t0
tm = H.timings()
runs = sum(r["timesRun"] for r in res)
print("mode=%s P=%d M=%d: fit %.2f s  (%.1f pairs/s, %.3g pair*SNP/s)  like kernels %.2f s  stage %.3f s  runs/pair %.2f  bad %d" % (
    a.mode, a.pairs, M, dt, a.pairs / dt, a.pairs * M / dt, tm["like_ms"] / 1e3, tm["stage_ms"] / 1e3, runs / a.pairs,
    sum(r["status"] != 0 for r in res)), flush=True)
nll = np.array([r["negloglik"] for r in res]); k1 = np.array([r["k1"] for r in res])
print("nll mean %.3f  k1 mean %.4f  k1>0.3: %d" % (nll.mean(), k1.mean(), (k1 > 0.3).sum()))
