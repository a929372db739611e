#!/usr/bin/env python3
"""Sensitivity of the reference's 2-parameter calcA fit (options.all.pair.txt = the mode of configs 2, 3 and 4) to
last-bit noise in its objective, per RESTART.

Run in the build container (needs oracle/_ref, i.e. /root/reference).  For the 190 keep-list pairs of dat_small with a
pinned seed: the reference's own sgl (emissions S, distances t, convInfo), then for each of the pair's 8 start points
(get_rand + get_start on the pinned rand() stream, src/relateHMM.cpp:278-293, 1546-1560) the reference's own
findmax_bfgs (src/bfgs.c:127-199 through the harness) with
  (0) the unperturbed objective (calcA_fun, src/relateHMM.cpp:1564-1595, restated by the C port, bit-identical to
      like() on the reference's own emissions) -- must reproduce the convInfo row of every run sgl consumed bit for bit;
  (k) K copies whose every objective value is moved by -1, 0 or +1 ulp at random (one stream per copy and pair), and
      K more whose values move by a whole number of ulps drawn uniformly from [-A, A], A chosen so that the noise has the
      root-mean-square size of the HIP objective's distance from like() -- measured on the device at the end points
      of these very runs by tools/objective_noise.py: rms 9.2 ulp on dat_small, 5.8 ulp on data/500 (two correct
      double-precision evaluations of the same likelihood, log-space there and scaled linear here, differ by that
      much) -> A = 16 (rms 9.5) and A = 10 (rms 6.1).  Keys "nll_copies" and "nll_copies_wide" ("amp_wide" = A).
The multi-start rule (run_optimization, :1852-1971) is a function of the 8 runs' -logL values alone; it is applied per
copy by tests/pathcheck.py:multistart.  What the fixture holds per pair: sgl's result, the 8 unperturbed rows
{a,k2,k1,-logL}, and per run the K perturbed -logL values (and k2, k1).  tests/test_gpu_sensitivity.py derives its bars
for the HIP path (objective equal to like() to ~1e-15, i.e. such a perturbation) from it.

A second fixture of the same form pins config 3 (PLINK data/500, all 66 pairs, same option file):
tests/golden/cli/c3_ref_fit.json (the reference's program prints neither -logL nor a).

Usage: make_sensitivity_allpair.py [workers] [c3|c2|all]    (~7 min on 7 workers)"""
import json
import multiprocessing as mp
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
OPTF = "options.all.pair.txt"
AMP_WIDE = {"dat_small": 16, "500": 10}


def work(args):
    """One worker = one reference session (the reference keeps static globals): pairs[lo:hi]."""
    data, lo, hi, copies = args
    ref = refload.Ref()
    o = oracleload.Oracle()
    g, pos, chr_ = pc.dataset(data)
    opt = dict(pc.options(OPTF))
    if data == "dat_small":
        pairs = [tuple(int(v) for v in p) for p in refload.keep_pairs(os.path.join(refload.DATA, "dat_small.keep"))]
    else:
        n = g.shape[0]
        pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    M = ref.init(g.astype(np.int32), pos, chr_, opt)
    R = opt["timesToRun"]
    stream = ref.rand_stream(SEED, 4000)
    starts = pc.draw_starts(stream, opt, R)
    out = []
    try:
        for n in range(lo, min(hi, len(pairs))):
            i, j = pairs[n]
            x = ref.sgl(i, j, seed=SEED)
            Sk = oracleload.Oracle.logs(x["S"])
            t = x["t"]

            def f0(v):
                return o.objective(1, v, opt["fixA_val"], opt["fixK_val"], opt["phi"], *Sk, t, opt["double_recom"],
                                   opt["alim1"])

            rows, nllc, nllw = [], [], []
            A = AMP_WIDE[data]
            for r in range(R):
                v, nf = ref.findmax_bfgs(f0, [starts[r][1], starts[r][2]], [0, 0], [1, 1])
                row = [ref.calculateA(1 - v[0] - v[1], v[1], v[0], opt["phi"]), float(v[0]), float(v[1]), float(-nf)]
                if r < x["timesRun"]:
                    assert np.array_equal(np.array(row), x["conv"][r]), (i, j, r, row, x["conv"][r])
                rows.append(row)
                for amp, dst in ((1, nllc), (A, nllw)):
                    a_ = []
                    for c in range(copies):
                        rng = np.random.default_rng(1000000 * c + 10 * n + r + 500000000 * (amp > 1))

                        def f(vv):
                            y = f0(vv)
                            s = int(rng.integers(-amp, amp + 1))
                            return float(y + s * np.spacing(y)) if s and np.isfinite(y) else y

                        vp, nfp = ref.findmax_bfgs(f, [starts[r][1], starts[r][2]], [0, 0], [1, 1])
                        a_.append(float(-nfp))
                    dst.append(a_)
            rec = {"pair": [i, j], "m": int(x["m"]), "a": float(x["a"]), "k2": float(x["k2"]), "k1": float(x["k1"]),
                   "nll": float(x["nll"]), "uLike": float(x["uLike"]), "poLike": float(x["poLike"]),
                   "timesRun": int(x["timesRun"]), "timesConv": int(x["timesConv"]), "runs": rows}
            if copies:
                rec.update(nll_copies=nllc, nll_copies_wide=nllw)
            out.append(rec)
            if (n - lo) % 5 == 0:
                print(data, n, i, j, x["nll"], x["timesRun"], flush=True)
    finally:
        ref.fini()
    return M, starts.tolist(), out


def run(data, npairs, copies, workers):
    step = (npairs + workers - 1) // workers
    jobs = [(data, lo, lo + step, copies) for lo in range(0, npairs, step)]
    with mp.Pool(len(jobs)) as pool:
        res = pool.map(work, jobs)
    M, starts = res[0][0], res[0][1]
    pairs = [p for r in res for p in r[2]]
    assert len(pairs) == npairs
    return {"options": OPTF, "data": data, "seed": SEED, "copies": copies, "amp_wide": AMP_WIDE[data], "M": int(M),
            "starts": starts, "pairs": pairs}


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    what = sys.argv[2] if len(sys.argv) > 2 else "all"
    if what in ("c3", "all"):
        c3 = run("500", 66, K, workers)
        with open(os.path.join(HERE, "cli", "c3_ref_fit.json"), "w") as fh:
            json.dump(c3, fh)
        print("c3_ref_fit.json (data/500, M %d)\n%s\n%s" % (c3["M"], pc.allpair_envelope_summary(c3),
                                                         pc.allpair_envelope_summary(c3, "nll_copies_wide")))
    if what in ("c2", "all"):
        out = run("dat_small", 190, K, workers)
        with open(os.path.join(HERE, "sensitivity_allpair.json"), "w") as fh:
            json.dump(out, fh)
        print("sensitivity_allpair.json (dat_small keep list, M %d)\n%s\n%s" % (
            out["M"], pc.allpair_envelope_summary(out), pc.allpair_envelope_summary(out, "nll_copies_wide")))


if __name__ == "__main__":
    main()
