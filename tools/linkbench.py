#!/usr/bin/env python3
"""Linkage statistic + permutation test (relate_hip_linkage = the reference's HMMtest) on the GPU against the CPU
oracle (plain-C port of src/HMMtest.txt, one core) on the same synthetic posterior matrix.
usage: python tools/linkbench.py [--ind 60] [--snps 20000] [--nsim 1000]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hmmtestcheck as hc
import relate_amd
ap = argparse.ArgumentParser()
ap.add_argument("--ind", type=int, default=60)
ap.add_argument("--snps", type=int, default=20000)
ap.add_argument("--nsim", type=int, default=1000)
ap.add_argument("--cpu-nsim", type=int, default=20)
a = ap.parse_args()
rng = np.random.default_rng(5)
n, M = a.ind, a.snps
R = n * (n - 1) // 2
post = rng.uniform(0, 1, size=(R, M)).astype(np.float32)
post[rng.random((R, M)) < 0.01] = -1
pos = np.cumsum(rng.uniform(0.01, 0.3, size=M)).astype(np.float32)
cc = (rng.random(n) < 0.5).astype(np.int32)
H = relate_amd.Relate(0)
H.linkage(post, cc, pos=pos, nsim=2, ccAll=2, infer=1, seed=1)           # warm-up (allocations, code load)
t0 = time.perf_counter()
got = H.linkage(post, cc, pos=pos, nsim=a.nsim, ccAll=2, infer=1, seed=1)
tg = time.perf_counter() - t0
inp = dict(post=post, pos=pos, K=None, isK0=0, cc=cc, nsim=a.cpu_nsim, ccAll=2, sig=0.05, infer=1)
t0 = time.perf_counter()
hc.oracle_linkage(inp, 1)
tc = time.perf_counter() - t0
inp["nsim"] = 2
t0 = time.perf_counter()
hc.oracle_linkage(inp, 1)
tc2 = time.perf_counter() - t0
per_perm = (tc - tc2) / (a.cpu_nsim - 2)
print("R=%d rows x M=%d SNPs: GPU %d permutations %.3f s (host buffers in and out); C port on one core: fixed part "
      "(inference + covariance) %.2f s, %.4f s per permutation -> %.1f s for %d permutations" % (
          R, M, a.nsim, tg, tc2 - 2 * per_perm, per_perm, tc2 + per_perm * (a.nsim - 2), a.nsim), flush=True)
