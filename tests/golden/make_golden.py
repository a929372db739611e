#!/usr/bin/env python3
"""Generates tests/golden/*.json from the compiled reference (oracle/_ref/librelate_ref.so, built from
/root/reference/src by oracle/Makefile).  Run here, in the build container; the JSON files are committed and
are what the GPU box checks against (the reference tree does not travel).

    python tests/golden/make_golden.py

Fixtures:
  ld_poly_coeffs.json  -- the 81 x 3 polynomials of LDemission_new, recovered by evaluating the reference with
                          unit weight vectors at random haplotype frequencies and solving for the integer
                          monomial coefficients
  golden_paths.json    -- per data set / option file: kept-SNP mask, LD cells, per-pair stage dumps (subsampled),
                          like() on a parameter grid, posteriors and Viterbi paths at fixed parameters, fitted
                          estimates + convInfo with a pinned seed, pruning keep mask, rand() stream, findmax on
                          analytic objectives
Floats are stored with repr() (exact round trip).
"""
import itertools
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import refload as R  # noqa: E402

SEED = 4242
GRID = [(0.05, 0.1, 0.3), (0.001, 0, 0), (0.001, 0, 1), (0.1, 0.5, 0.5), (0.149, 0.0, 0.25), (0.2, 0.1, 0.1),
        (0.1, 0.7, 0.4), (0.0905, 0.0, 0.1124), (0.15, 0.25, 0.5), (0.01, 1.0, 0.0), (0.03, 0.0, 0.999),
        (0.12, -0.1, 0.5), (0.08, 0.3, 0.0), (0.0010000000474974513, 0.2, 0.8)]
DECODE_PARS = [(0.05, 0.2, 0.3), (0.0905, 0.0, 0.1124), (0.02, 0.0, 1.0)]


def fl(a):
    return [float(x) for x in np.asarray(a, dtype=np.float64).ravel()]


def poly_coeffs(ref):
    rng = np.random.default_rng(1)
    n = 60
    X = rng.uniform(0.1, 1.0, size=(n, 4))  # pBA, pbA, pBa, pba
    out = {}
    for c in range(81):
        ep = np.zeros((81, n))
        ep[c] = 1
        val = ref.ldemission(X[:, 0], X[:, 1], X[:, 2], X[:, 3], ep)
        for row, deg in ((0, 4), (1, 3), (2, 2)):
            ms = [e for e in itertools.product(range(deg + 1), repeat=4) if sum(e) == deg]
            A = np.array([[np.prod(x ** np.array(e)) for e in ms] for x in X])
            coef, *_ = np.linalg.lstsq(A, val[row], rcond=None)
            ci = np.round(coef).astype(int)
            assert np.abs(coef - ci).max() < 1e-6
            out["%d,%d" % (c, row)] = [[int(k), list(e)] for k, e in zip(ci, ms) if k != 0]
    return out


def dataset_block(ref, name, g, pos, chr_, optf, pairs, seeds=(SEED,)):
    opt = R.read_options(os.path.join(R.DATA, optf))
    M = ref.init(g, pos, chr_, opt)
    keep, genos, maf, posg = ref.globals()
    blk = {"data": name, "options": optf, "M": M, "keep_idx": [int(i) for i in np.where(keep == 1)[0]],
           "maf_sub": fl(maf[::37]), "pairs": []}
    if opt["ld_adj"] and opt["back"] > 0:
        hap = ref.hap_global()
        cells = []
        rng = np.random.default_rng(5)
        for _ in range(120):
            d = int(rng.integers(0, opt["back2"]))
            s = int(rng.integers(d + 1, M))
            cells.append([d, s, float(hap["mea"][d, s - 1]), float(hap["pBA"][d, s]), float(hap["pBa"][d, s]),
                          float(hap["pbA"][d, s]), float(hap["pba"][d, s])])
        blk["hap_cells"] = cells
    for (i, j) in pairs:
        x = ref.sgl(i, j, seed=seeds[0])
        m = x["m"]
        with np.errstate(divide="ignore"):
            Sk = (np.ascontiguousarray(np.log(x["S"][2])), np.ascontiguousarray(np.log(x["S"][1])),
                  np.ascontiguousarray(np.log(x["S"][0])))
        sub = list(range(0, m, max(1, m // 40)))
        rec = {"pair": [i, j], "m": m, "dropped_idx": [int(s) for s in np.where(x["keep"] == 0)[0]],
               "sub": sub, "t_sub": fl(x["t"][sub]), "S_sub": [fl(x["S"][r][sub]) for r in range(3)],
               "S1_sub": [fl(x["S1"][r][sub]) for r in range(3)], "choose_sub": [int(x["choose"][s]) for s in sub],
               "like_grid": [float(ref.like(d, *Sk, x["t"], opt["double_recom"], [opt["alim0"], opt["alim1"]]))
                             for d in GRID],
               "decode": [], "fit": {k: (float(x[k]) if k not in ("timesRun", "timesConv") else int(x[k]))
                                     for k in ("a", "k2", "k1", "k0", "nll", "uLike", "poLike", "timesRun",
                                               "timesConv")},
               "conv": [fl(r) for r in x["conv"][:x["timesRun"]]]}
        for par in DECODE_PARS:
            post = ref.decode(par, *Sk, x["t"], opt["double_recom"])
            path = ref.viterbi(par, *Sk, x["t"], opt["double_recom"])
            rec["decode"].append({"par": list(par), "post_sub": [fl(post[r][sub]) for r in range(3)],
                                  "path": "".join(str(int(v)) for v in path)})
        # posterior and path at the fitted estimate, as sgl returns them
        rec["post_fit_sub"] = [fl(x["post"][r][sub]) for r in range(3)]
        rec["path_fit"] = "".join(str(int(v)) for v in x["path"])
        blk["pairs"].append(rec)
    ref.fini()
    return blk


def main():
    ref = R.Ref()
    with open(os.path.join(HERE, "ld_poly_coeffs.json"), "w") as fh:
        json.dump(poly_coeffs(ref), fh, separators=(",", ":"))
    g, pos, chr_ = R.load_dataset("dat_small")
    gp, posp, chrp = R.load_plink(os.path.join(R.DATA, "500"))
    out = {"seed": SEED, "grid": [list(d) for d in GRID], "blocks": []}
    out["blocks"].append(dataset_block(ref, "dat_small", g, pos, chr_, "options.single.pair.txt",
                                       [(0, 1), (5, 12), (2, 9), (3, 50), (100, 103)]))
    out["blocks"].append(dataset_block(ref, "dat_small", g, pos, chr_, "options.all.pair.txt",
                                       [(0, 1), (2, 9), (7, 8)]))
    out["blocks"].append(dataset_block(ref, "dat_small", g, pos, chr_, "options.linkage.txt",
                                       [(0, 1), (5, 12), (11, 19), (4, 17)]))
    out["blocks"].append(dataset_block(ref, "500", gp, posp, chrp, "options.all.pair.txt",
                                       [(0, 1), (3, 7), (10, 11), (2, 5)]))
    out["rand_stream"] = {"seed": SEED, "values": [int(v) for v in ref.rand_stream(SEED, 40)]}

    # findmax_bfgs on analytic objectives (pins the L-BFGS-B restatement without needing the .so on the GPU box)
    def f1(x):
        return (x[0] - 0.3) ** 2 * 7 + (x[1] - 0.8) ** 4 + 0.5 * x[0] * x[1] + 3

    def rosen(x):
        return 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2 + (x[2] - 0.2) ** 2

    def f3(x):
        return float(np.exp(x[0]) - 2 * x[0])

    fm = []
    for name, f, lo, hi, starts in (("f1", f1, [0, 0], [1, 1], [[0.1, 0.2], [0.9, 0.95], [0.0, 1.0], [0.5, 0.5]]),
                                    ("rosen", rosen, [-1, -1, 0], [0.8, 2, 1],
                                     [[0.1, 0.2, 0.5], [0.9, 0.1, 0.9], [-0.5, 0.5, 0.0]]),
                                    ("f3", f3, [0], [1], [[0.1], [0.99], [0.0]])):
        for x0 in starts:
            x, r = ref.findmax_bfgs(f, x0, lo, hi)
            fm.append({"fun": name, "x0": x0, "lo": lo, "hi": hi, "x": fl(x), "ret": float(r)})
    out["findmax"] = fm
    with open(os.path.join(HERE, "golden_paths.json"), "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", os.path.getsize(os.path.join(HERE, "golden_paths.json")), "bytes")


if __name__ == "__main__":
    main()
