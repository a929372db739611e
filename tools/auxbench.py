#!/usr/bin/env python3
"""The kernels bench.py's step does not touch, at sizes worth profiling: the Viterbi (relate_hip_decode with a path,
config-4-length rows) and the linkage kernels (relate_hip_linkage, 1 770 rows x 20 000 SNPs, 1 000 permutations).
Run under tools/trace_tool.sh / tools/pmc_run.sh for the per-kernel numbers.
usage: python tools/auxbench.py [--pairs 2048] [--snps 500000] [--inds 2000] [--skip-viterbi] [--skip-linkage]"""
import argparse
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import relate_amd  # noqa: E402
from relate_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=2048)
ap.add_argument("--snps", type=int, default=500000)
ap.add_argument("--inds", type=int, default=2000)
ap.add_argument("--link-ind", type=int, default=60)
ap.add_argument("--link-snps", type=int, default=20000)
ap.add_argument("--nsim", type=int, default=1000)
ap.add_argument("--skip-viterbi", action="store_true")
ap.add_argument("--skip-linkage", action="store_true")
a = ap.parse_args()
H = relate_amd.Relate(0)
if not a.skip_viterbi:
    opt = bench.opts_all_pair()
    opt["back2"] = 2 * opt["back"]
    geno, pos, chr_, _ = synth.make_dataset(a.inds, a.snps, device="cuda:0")
    M = H.init_globals(geno, pos, chr_, opt)
    blk = synth.all_pairs(a.inds)[:a.pairs]
    par = np.ascontiguousarray(bench.eval_points(a.pairs, 1)[:, 0, :])
    H.decode(blk[:64], par[:64], want_post=False, want_path=True)            # warm-up
    t0 = time.perf_counter()
    _, path, ns = H.decode(blk, par, want_post=False, want_path=True)
    dt = time.perf_counter() - t0
    print("viterbi: %d pairs x %d SNPs in %.3f s including staging (full-format records) and the path copy: "
          "%.2e pair-SNPs/s; states %s" % (a.pairs, M, dt, a.pairs * M / dt, np.bincount(path.ravel() + 1, minlength=4)),
          flush=True)
    del path
if not a.skip_linkage:
    rng = np.random.default_rng(5)
    n, Ml = a.link_ind, a.link_snps
    R = n * (n - 1) // 2
    post = rng.uniform(0, 1, size=(R, Ml)).astype(np.float32)
    post[rng.random((R, Ml)) < 0.01] = -1
    lpos = np.cumsum(rng.uniform(0.01, 0.3, size=Ml)).astype(np.float32)
    cc = (rng.random(n) < 0.5).astype(np.int32)
    H.linkage(post, cc, pos=lpos, nsim=2, ccAll=2, infer=1, seed=1)          # warm-up
    t0 = time.perf_counter()
    H.linkage(post, cc, pos=lpos, nsim=a.nsim, ccAll=2, infer=1, seed=1)
    dt = time.perf_counter() - t0
    print("linkage: %d rows x %d SNPs, %d permutations in %.3f s (host buffers in and out)" % (R, Ml, a.nsim, dt),
          flush=True)
