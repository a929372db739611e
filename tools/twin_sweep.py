#!/usr/bin/env python3
"""Ad-hoc sweep on the GPU box: per-pair emissions, kept SNPs, distances and the two fixed-point likelihoods (uLike,
poLike) of the HIP path against the compiled reference (oracle/_ref, init_globals + sgl) over synthetic data sets and
option variations the golden fixtures do not cover.  Prints one line per configuration; exits 1 on a mismatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pathcheck as pc, refload
import relate_amd
from relate_amd import synth

ref = refload.Ref()
H = relate_amd.Relate(0)
base = dict(pc.options("options.single.pair.txt"))
CASES = [
    ("N12 M2000", dict(N=12, M=2000), {}),
    ("N50 M2500 miss 3%", dict(N=50, M=2500, miss=0.03), {}),
    # (back = 1 is excluded from parity: remove_cross zeroes a row at matrix depth 2 instead of shifting)
    ("N24 M3000 back 2", dict(N=24, M=3000), dict(back=2, back2=4)),
    ("N24 M3000 back 12", dict(N=24, M=3000, miss=0.01), dict(back=12, back2=24)),
    ("N24 M3000 back 20 (generic staging)", dict(N=24, M=3000, miss=0.01), dict(back=20, back2=40)),
    ("N30 M3000 LD=0", dict(N=30, M=3000), dict(LD=0)),
    # (ld_adj = 0 is covered against the oracle in tests/test_gpu_parity.py: the reference's sgl segfaults on it here)
    ("N30 M3000 double_recom", dict(N=30, M=3000), dict(double_recom=1)),
    ("N30 M3000 eps 0.05 min 0.1", dict(N=30, M=3000), dict(epsilon=float(np.float32(0.05)), min=float(np.float32(0.1)))),
    ("N40 M4000 prune", dict(N=40, M=4000), dict(doPrune=1, prune_val=float(np.float32(0.1)), back=10, back2=20)),
    ("N16 M1500 copy 0.6", dict(N=16, M=1500, copy_p=0.6), {}),
]
bad = 0
for name, dk, ok in CASES:
    kw = dict(dk); N = kw.pop("N"); M = kw.pop("M")
    g, pos, chr_, related = synth.make_dataset(N, M, n_related=3, seed=20240601 + len(name), **kw)
    opt = dict(base, **ok)
    opt.update(doPar=1, par=[0.05, 0.1, 0.3])   # no optimisation: likelihood, posteriors and path at (a, k2, k1)
    Mref = ref.init(g.astype(np.int32), pos, chr_, opt)
    try:
        H.init_globals(g, pos, chr_, dict(opt, seed=1, seed_mode=0))
        pairs = [(int(r[0]), int(r[1])) for r in related] + [(0, N - 1), (1, N // 2)]
        neq = 0.0
        if opt["ld_adj"]:
            hg, hr = H.hap_global(), ref.hap_global()
            neq = max(float((hg[k] != hr[k]).mean()) for k in ("mea", "pBA", "pBa", "pbA", "pba"))
        worst = 0.0; ok_keep = True; ok_t = True; lk = 0.0
        res, post, path = H.sgl(pairs, want_path=True)   # doPar: likelihood, posteriors and path at par
        pw = 0.0; pneq = 0; npath = 0
        for n, (i, j) in enumerate(pairs):
            x = ref.sgl(i, j, seed=1)
            kp, tt, S = H.pair_emissions(i, j)
            idx = np.where(kp == 1)[0]
            ok_keep &= np.array_equal(kp, x["keep"])
            ok_t &= np.array_equal(tt[idx][1:], x["t"][1:x["m"]])
            if len(idx) == x["m"]:
                worst = max(worst, float(np.abs(S[:, idx] / x["S"] - 1).max()))
            lk = max(lk, abs(res[n]["uLike"] / x["uLike"] - 1), abs(res[n]["poLike"] / x["poLike"] - 1),
                     abs(res[n]["negloglik"] / x["nll"] - 1))
            if len(idx) == x["m"]:
                pw = max(pw, float(np.abs(post[n][:, idx] - x["post"]).max()))
                pneq += int((path[n][idx] != x["path"]).sum()); npath += len(idx)
        flag = "" if (H.M == Mref and ok_keep and ok_t and worst < 1e-9 and lk < 1e-10 and pw < 1e-8 and pneq <= 0.002 * npath) else "   <-- MISMATCH"
        bad += bool(flag)
        print("%-38s M %d/%d keep %s t %s  S maxrel %.2e  like (u, po, par) maxrel %.2e  post maxabs %.1e  path neq %d/%d  LD cells differing %.3f %%%s" % (
            name, H.M, Mref, ok_keep, ok_t, worst, lk, pw, pneq, npath, 100 * neq, flag), flush=True)
    finally:
        ref.fini()
# ---- the optimisation modes (objective wrappers) with the multi-start fit, pinned seed ---------------------------
MODES = [
    ("3 free (a, k2, k1)", dict(calcA=0, fixA=0, fixK=0)),
    ("fixK2=0: a, k1 free", dict(calcA=0, fixA=0, fixK=1, fixK_val=0.0)),
    ("fixK2=0 + calcA: k1 free", dict(calcA=1, fixA=0, fixK=1, fixK_val=0.0)),
    ("fixK2=0.1: a, k1 free", dict(calcA=0, fixA=0, fixK=1, fixK_val=float(np.float32(0.1)))),
]
N, M, seed = 24, 3000, 777
g, pos, chr_, related = synth.make_dataset(N, M, n_related=4, miss=0.005)
pairs = [(int(r[0]), int(r[1])) for r in related] + [(0, N - 1), (1, N // 2)]
for name, ok in MODES:
    opt = dict(base, **ok)
    Mref = ref.init(g.astype(np.int32), pos, chr_, opt)
    try:
        H.init_globals(g, pos, chr_, dict(opt, seed=seed, seed_mode=0))
        res, post, _ = H.sgl(pairs)
        on = 0; worst = 0.0; kd = 0.0; pw = 0.0; never_worse = True; obj = 0.0; detail = []
        for n, (i, j) in enumerate(pairs):
            x, r = ref.sgl(i, j, seed=seed), res[n]
            rel = abs(r["negloglik"] / x["nll"] - 1)
            # the objective itself, at the point where the reference stopped
            at_ref = H.like([(i, j)], np.array([[[x["a"], x["k2"], x["k1"]]]]))[0, 0]
            obj = max(obj, abs(at_ref / x["nll"] - 1))
            if rel > 1e-9:
                detail.append("      (%d,%d) -logL %.10g / ref %.10g   k1 %.6g / %.6g   a %.5g / %.5g   runs %d / %d" % (
                    i, j, r["negloglik"], x["nll"], r["k1"], x["k1"], r["a"], x["a"], r["timesRun"], x["timesRun"]))
            worst = max(worst, rel)
            never_worse &= r["negloglik"] <= x["nll"] * (1 + 1e-6)
            if rel <= 1e-9:
                on += 1
                kd = max(kd, abs(r["k1"] - x["k1"]), abs(r["k2"] - x["k2"]), abs(r["a"] - x["a"]))
                idx = np.where(x["keep"] == 1)[0]
                pw = max(pw, float(np.abs(post[n][:, idx] - x["post"]).max()))
        flag = "" if (obj < 1e-11 and kd < 1e-6 and pw < 1e-6) else "   <-- MISMATCH"
        bad += bool(flag)
        print("fit %-28s objective at the reference's estimate maxrel %.1e; -logL reached maxrel %.2e, never worse %s; on the "
              "reference's trajectory %d/%d (estimates maxabs %.1e, posteriors %.1e)%s" % (
                  name, obj, worst, never_worse, on, len(pairs), kd, pw, flag), flush=True)
        for d in detail:
            print(d, flush=True)
    finally:
        ref.fini()
sys.exit(1 if bad else 0)
