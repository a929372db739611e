#!/usr/bin/env python3
"""All 190 keep-list pairs of dat_small: the HIP path's multi-start fit against the compiled reference run live with the
same pinned seed (oracle/_ref: init_globals + sgl), in the 3-parameter mode (options.single.pair.txt) and the
1-parameter pruned linkage mode (options.linkage.txt = config 5).  Prints how many pairs follow the reference's
trajectory and the worst differences.  GPU box only (the reference runs on its host cores)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pathcheck as pc, refload
import relate_amd

ref = refload.Ref()
H = relate_amd.Relate(0)
g, pos, chr_ = refload.load_dataset("dat_small")
pairs = [tuple(int(v) for v in p) for p in refload.keep_pairs(os.path.join(refload.DATA, "dat_small.keep"))]
seed = 4242
for optf in ("options.single.pair.txt", "options.linkage.txt"):
    opt = dict(pc.options(optf))
    Mref = ref.init(g, pos, chr_, opt)
    try:
        H.init_globals(g, pos, chr_, dict(opt, seed=seed, seed_mode=0))
        t0 = time.perf_counter()
        res, post, path = H.sgl(pairs, want_path=True)
        tg = time.perf_counter() - t0
        on = 0; rel = []; better = 0; worse = 0; kd = 0.0; pw = 0.0; pneq = 0; npath = 0; offk = 0.0; t_ref = 0.0
        for n, (i, j) in enumerate(pairs):
            t0 = time.perf_counter()
            x = ref.sgl(i, j, seed=seed)
            t_ref += time.perf_counter() - t0
            r = res[n]
            d = r["negloglik"] / x["nll"] - 1
            rel.append(abs(d))
            if abs(d) <= 1e-9:
                on += 1
                kd = max(kd, abs(r["k1"] - x["k1"]), abs(r["k2"] - x["k2"]), abs(r["a"] - x["a"]))
                idx = np.where(x["keep"] == 1)[0]
                pw = max(pw, float(np.abs(post[n][:, idx] - x["post"]).max()))
                pneq += int((path[n][idx] != x["path"]).sum()); npath += len(idx)
            else:
                better += d < 0; worse += d > 0
                offk = max(offk, abs(r["k1"] - x["k1"]))
        print("%s: M %d/%d, %d pairs; GPU sgl %.2f s, reference %.1f s on one core" % (optf, H.M, Mref, len(pairs), tg, t_ref))
        print("   on the reference's trajectory (-logL within 1e-9): %d/%d; there estimates maxabs %.1e, posteriors maxabs %.1e, "
              "Viterbi states differing %d/%d" % (on, len(pairs), kd, pw, pneq, npath))
        print("   off it: %d with lower -logL, %d with higher; -logL maxrel %.2e, k1 maxabs %.2e" % (better, worse, max(rel), offk),
              flush=True)
    finally:
        ref.fini()
