#!/usr/bin/env python3
"""Sensitivity of the reference's 1-parameter fit (options.linkage.txt = config 5) to last-bit noise in its objective.

Run in the build container (needs oracle/_ref, i.e. /root/reference).  For the 190 keep-list pairs of dat_small, pinned
seed: the reference's own sgl (emissions S, distances t, estimate), then the reference's own findmax_bfgs
(src/bfgs.c:127-199 through the harness) on the same start point with
  (0) the unperturbed objective -- must reproduce sgl's estimate bit for bit (the wrapper restated here is
      fixK2_calcA_fun, src/relateHMM.cpp:1736-1781; the objective is the C port's, bit-identical to like() on the
      reference's own emissions, tests/test_fit_cpu.py);
  (k) K copies of it whose every value is moved by -1, 0 or +1 ulp at random (a different stream per copy).
The spread of the fitted -logL and k1 over the copies is how far the reference's result moves when its objective is
equal to ~1e-16 but not bit-equal: the envelope the HIP path's fit (objective equal to the reference's to ~1e-15) is
held to in tests/test_gpu_sensitivity.py.  Output: tests/golden/sensitivity_linkage.json."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracleload  # noqa: E402
import pathcheck as pc  # noqa: E402
import refload  # noqa: E402

SEED, K = 4242, 16
OPTF = "options.linkage.txt"


def main():
    ref = refload.Ref()
    o = oracleload.Oracle()
    g, pos, chr_ = refload.load_dataset("dat_small")
    opt = dict(pc.options(OPTF))
    pairs = [tuple(int(v) for v in p) for p in refload.keep_pairs(os.path.join(refload.DATA, "dat_small.keep"))]
    M = ref.init(g, pos, chr_, opt)
    stream = ref.rand_stream(SEED, 64)
    k1_start = pc.draw_starts(stream, opt, 1)[0][2]
    fk = opt["fixK_val"]
    lb = -2 * fk + 0.5 * np.sqrt(32 * fk - 16 * fk)
    out = {"options": OPTF, "seed": SEED, "copies": K, "M": int(M), "k1_start": float(k1_start), "pairs": []}
    try:
        for n, (i, j) in enumerate(pairs):
            x = ref.sgl(i, j, seed=SEED)
            Sk = oracleload.Oracle.logs(x["S"])
            t = x["t"]

            def f0(v):
                return o.objective(3, v, opt["fixA_val"], fk, opt["phi"], *Sk, t, opt["double_recom"], opt["alim1"])

            xa, nf = ref.findmax_bfgs(f0, [k1_start], [lb], [1])
            assert -nf == x["nll"] and xa[0] == x["k1"], (i, j, -nf, x["nll"], xa[0], x["k1"])
            nll, k1, nev = [], [], []
            for c in range(K):
                rng = np.random.default_rng(100000 * c + n)
                cnt = [0]

                def f(v):
                    cnt[0] += 1
                    y = f0(v)
                    s = int(rng.integers(-1, 2))
                    return float(np.nextafter(y, np.inf if s > 0 else -np.inf)) if s else y

                xp, nfp = ref.findmax_bfgs(f, [k1_start], [lb], [1])
                nll.append(float(-nfp))
                k1.append(float(xp[0]))
                nev.append(cnt[0])
            out["pairs"].append({"pair": [i, j], "nll": float(x["nll"]), "k1": float(x["k1"]), "a": float(x["a"]),
                                 "nll_copies": nll, "k1_copies": k1, "evals": nev})
            if n % 20 == 0:
                print(n, i, j, x["nll"], min(nll) - x["nll"], max(nll) - x["nll"], flush=True)
    finally:
        ref.fini()
    with open(os.path.join(HERE, "sensitivity_linkage.json"), "w") as fh:
        json.dump(out, fh)
    # summary
    rel = np.array([[c / p["nll"] - 1 for c in p["nll_copies"]] for p in out["pairs"]])
    dk = np.array([[c - p["k1"] for c in p["k1_copies"]] for p in out["pairs"]])
    print("pairs %d, copies %d" % rel.shape)
    print("on the unperturbed trajectory (|rel| <= 1e-9): per copy min %d max %d of %d" % (
        (np.abs(rel) <= 1e-9).sum(axis=0).min(), (np.abs(rel) <= 1e-9).sum(axis=0).max(), rel.shape[0]))
    print("worse than unperturbed (rel > 1e-9): per copy min %d max %d" % ((rel > 1e-9).sum(axis=0).min(), (rel > 1e-9).sum(axis=0).max()))
    print("max rel excess %.3e, max rel gain %.3e, max |dk1| %.3e" % (rel.max(), -rel.min(), np.abs(dk).max()))


if __name__ == "__main__":
    main()
