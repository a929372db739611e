"""Shared helpers of the parity tests: data set loading, the oracle's pre_calc, golden fixture access."""
import json
import os

import numpy as np

import refload as R

GOLDEN = os.path.join(R.ROOT, "tests", "golden")
RAND_MAX = 2147483647.0

_cache = {}


def golden():
    if "g" not in _cache:
        with open(os.path.join(GOLDEN, "golden_paths.json")) as fh:
            _cache["g"] = json.load(fh)
    return _cache["g"]


def poly_golden():
    with open(os.path.join(GOLDEN, "ld_poly_coeffs.json")) as fh:
        return json.load(fh)


def dataset(name):
    if name not in _cache:
        if name == "500":
            _cache[name] = R.load_plink(os.path.join(R.DATA, "500"))
        else:
            _cache[name] = R.load_dataset(name)
    return _cache[name]


def options(optf):
    return R.read_options(os.path.join(R.DATA, optf))


def oracle_precalc(o, g, pos, opt):
    """pre_calc (src/relateHMM.cpp:3293-3362) with the oracle's pieces."""
    g8 = np.ascontiguousarray(g, dtype=np.uint8)
    maf0 = o.maf(g8)
    keep = o.maf_keep(maf0, opt["min"])
    gs = np.ascontiguousarray(g8[:, keep == 1])
    if opt["doPrune"]:
        k2 = o.prune(gs, opt["LD"], opt["back"], opt["prune_val"])
        full = keep.copy()
        full[np.where(keep == 1)[0][k2 == 0]] = 0
        keep = full
        gs = np.ascontiguousarray(g8[:, keep == 1])
    ps = np.ascontiguousarray(np.asarray(pos)[keep == 1])
    mafs = o.maf(gs)
    use_ld = 1 if (opt["ld_adj"] and opt["back"] > 0) else 0
    hap = o.hap_global(gs, opt["back2"], opt["LD"]) if use_ld else None
    return dict(keep=keep, gs=gs, ps=ps, mafs=mafs, hap=hap, use_ld=use_ld)


def oracle_stage(o, pc, opt, i, j):
    st = o.pair_stage(pc["gs"], i, j, pc["mafs"], pc["ps"], opt["epsilon"], pc["use_ld"], opt["back"], opt["back2"],
                      pc["hap"])
    st["Sk"] = o.logs(st["S"])
    return st


def fit_mode(opt):
    if opt["fixK"] and opt["calcA"] and not opt["fixA"]:
        return 3
    if opt["fixK"] and opt["fixA"] and not opt["calcA"]:
        return 4
    if opt["fixK"] and not opt["fixA"] and not opt["calcA"]:
        return 2
    if not opt["fixK"] and not opt["fixA"] and not opt["calcA"]:
        return 0
    if not opt["fixK"] and not opt["fixA"] and opt["calcA"]:
        return 1
    raise ValueError("unsupported mode")


def draw_starts(stream, opt, nruns):
    """get_rand / get_start (src/relateHMM.cpp:278-293, 1546-1560) on an explicit rand() stream;
    returns nruns x 4 start records {a,k2,k1,unused}."""
    mode = fit_mode(opt)
    st = np.asarray(stream, dtype=np.float64)
    at = 0
    out = np.zeros((nruns, 4))
    for r in range(nruns):
        a = opt["alim0"] + opt["alim1"] * (st[at] / RAND_MAX)
        at += 1
        if opt["fixK"]:
            k2 = opt["fixK_val"]
        else:
            k2 = st[at] / RAND_MAX
            at += 1
        k1 = (1 - k2) * st[at] / RAND_MAX
        at += 2
        if mode == 1:
            while True:
                k2 = st[at] / RAND_MAX
                k1 = ((1 - k2) * st[at + 1]) / RAND_MAX
                at += 2
                if not ((k1 + 2 * k2) ** 2 < 4 * k2):
                    break
        out[r] = (a, k2, k1, 0.0)
    return out


def multistart(nll, timesToConverge, timesToRun, convTol):
    """run_optimization's restart rule (src/relateHMM.cpp:1909-1960) on the -logL of a pair's runs in start order:
    returns (index of the best run, timesRun, timesConverged)."""
    best, conv, run = 0, 1, 1
    while conv < timesToConverge and run < timesToRun:
        v = nll[run]
        run += 1
        if abs(v - nll[best]) < convTol:
            conv += 1
            continue
        if abs(v) < abs(nll[best]):
            best, conv = run - 1, 1
    return best, run, conv


def allpair_envelope(fx, opt, key="nll_copies"):
    """Per-copy statistics of tests/golden/sensitivity_allpair.json (tests/golden/make_sensitivity_allpair.py): what
    the reference's 2-parameter calcA fit does when every objective value carries -1/0/+1 ulp of noise.
      ref_nll[p][r]      unperturbed -logL of run r (rows r < timesRun are sgl's convInfo)
      rel[p][r][c]       perturbed / unperturbed - 1
      on[c]              (pair, run) results of copy c within 1e-9 relative of the unperturbed run
      best_excess[p][c]  best-of-runs -logL of copy c (the restart rule applied to the copy) / sgl's -logL - 1
      runs[c], conv[c]   the copy's timesRun / timesConverged per pair"""
    ref_nll = np.array([[r[3] for r in p["runs"]] for p in fx["pairs"]])
    cop = np.array([p[key] for p in fx["pairs"]])                     # pairs x runs x copies
    rel = cop / ref_nll[:, :, None] - 1
    P, R, K = cop.shape
    best_excess = np.zeros((P, K))
    runs = np.zeros((K, P), dtype=int)
    conv = np.zeros((K, P), dtype=int)
    for p in range(P):
        for c in range(K):
            b, nr, nc = multistart(cop[p, :, c], opt["timesToConverge"], opt["timesToRun"], opt["convTol"])
            best_excess[p, c] = cop[p, b, c] / fx["pairs"][p]["nll"] - 1
            runs[c, p], conv[c, p] = nr, nc
    return dict(ref_nll=ref_nll, rel=rel, on=(np.abs(rel) <= 1e-9).sum(axis=(0, 1)), best_excess=best_excess,
                runs=runs, conv=conv)


def allpair_envelope_summary(fx, key="nll_copies"):
    opt = options(fx["options"])
    e = allpair_envelope(fx, opt, key)
    P, R, K = e["rel"].shape
    return ("noise amplitude %d ulp: pairs %d x runs %d x copies %d\n(pair, run) results on the unperturbed trajectory (|rel| <= 1e-9): per copy "
            "min %d max %d of %d\nrun -logL: max rel excess %.3e, max rel gain %.3e\nbest-of-runs -logL vs sgl's: max "
            "excess %.3e, max gain %.3e, pairs whose best moved by > 1e-9: per copy min %d max %d\ntimesRun per copy: "
            "mean %.3f..%.3f; timesConverged > 1: %d..%d pairs" % (
                1 if key == "nll_copies" else fx["amp_wide"], P, R, K, e["on"].min(), e["on"].max(), P * R, e["rel"].max(), -e["rel"].min(),
                e["best_excess"].max(), -e["best_excess"].min(),
                (np.abs(e["best_excess"]) > 1e-9).sum(axis=0).min(), (np.abs(e["best_excess"]) > 1e-9).sum(axis=0).max(),
                e["runs"].mean(axis=1).min(), e["runs"].mean(axis=1).max(),
                (e["conv"] > 1).sum(axis=1).min(), (e["conv"] > 1).sum(axis=1).max()))


def allpair_bars(fx, opt, res, conv, key="nll_copies_wide"):
    """The HIP fit of the fixture's pairs (res = relate_fit records, conv = P x timesToRun x 4 convInfo) against the
    envelope of the fixture's perturbed copies -- by default the copies whose noise has the measured size of the HIP
    objective's distance from like() ("nll_copies_wide"; "nll_copies" = +-1 ulp).  Like is compared with like: a fit only reports the runs its restart
    rule consumed, so a copy's runs count up to the copy's own timesRun.  Returns (stats, failures); the bars:
      (i)   (pair, run) results on the reference's trajectory (-logL within 1e-9 relative of the unperturbed run):
            at least the worst copy's count less max(10, 3 standard deviations of the copies' counts);
      (ii)  best-of-runs -logL relative to sgl's: excess at most the copies' largest, gain at most twice their largest;
      (iii) mean timesRun and the number of pairs with timesConverged > 1: inside the copies' range widened by two
            of their standard deviations (16 copies are a sample, not a bound).
    On-trajectory runs must also carry the reference's (a, k2, k1) to 1e-6."""
    e = allpair_envelope(fx, opt, key)
    P, R, K = e["rel"].shape
    on_c = np.array([sum(int((np.abs(e["rel"][p, :e["runs"][c, p], c]) <= 1e-9).sum()) for p in range(P)) for c in range(K)])
    used_c = e["runs"].sum(axis=1)
    nll = np.array([r["negloglik"] for r in res])
    ref = np.array([p["nll"] for p in fx["pairs"]])
    tr = np.array([r["timesRun"] for r in res])
    tc = np.array([r["timesConverged"] for r in res])
    fails = []
    on = 0
    for p in range(P):
        for r in range(tr[p]):
            want = fx["pairs"][p]["runs"][r]
            if abs(conv[p][r][3] / want[3] - 1) <= 1e-9:
                on += 1
                if not np.allclose(conv[p][r][:3], want[:3], rtol=1e-6, atol=1e-7):
                    fails.append("pair %s run %d: on trajectory by -logL but estimates differ" % (fx["pairs"][p]["pair"], r))
    d = nll / ref - 1
    margin = max(10.0, 3 * on_c.std())
    moved = (np.abs(e["best_excess"]) > 1e-9).sum(axis=0)
    multi = (e["conv"] > 1).sum(axis=1)
    mr = e["runs"].mean(axis=1)
    st = dict(on=on, used=int(tr.sum()), on_copies=(int(on_c.min()), int(on_c.max())), used_copies=(int(used_c.min()), int(used_c.max())),
              margin=margin, excess=float(d.max()), gain=float(-d.min()), excess_copies=float(e["best_excess"].max()),
              gain_copies=float(-e["best_excess"].min()), moved=int((np.abs(d) > 1e-9).sum()), moved_copies=(int(moved.min()), int(moved.max())),
              mean_runs=float(tr.mean()), mean_runs_copies=(float(mr.min()), float(mr.max())),
              multi=int((tc > 1).sum()), multi_copies=(int(multi.min()), int(multi.max())),
              mean_runs_ref=float(np.mean([p["timesRun"] for p in fx["pairs"]])),
              multi_ref=int(sum(p["timesConv"] > 1 for p in fx["pairs"])))
    if on < on_c.min() - margin:
        fails.append("(i) on trajectory %d < worst copy %d - %.0f" % (on, on_c.min(), margin))
    if d.max() > e["best_excess"].max():
        fails.append("(ii) excess %.3e > copies' %.3e" % (d.max(), e["best_excess"].max()))
    if -d.min() > 2 * -e["best_excess"].min():
        fails.append("(ii) gain %.3e > twice the copies' %.3e" % (-d.min(), -e["best_excess"].min()))
    if st["moved"] > moved.max() + 2 * moved.std() + 2:
        fails.append("(ii) pairs whose best moved %d > copies' %d" % (st["moved"], moved.max()))
    if not (mr.min() - 2 * mr.std() - 0.02 <= tr.mean() <= mr.max() + 2 * mr.std() + 0.02):
        fails.append("(iii) mean timesRun %.3f outside %.3f..%.3f" % (tr.mean(), mr.min(), mr.max()))
    if not (multi.min() - 2 * multi.std() - 2 <= st["multi"] <= multi.max() + 2 * multi.std() + 2):
        fails.append("(iii) pairs with timesConverged > 1: %d outside %d..%d" % (st["multi"], multi.min(), multi.max()))
    return st, fails


def allpair_line(name, st):
    return ("%s: (pair, run) on the reference's trajectory %d of %d consumed (copies %d..%d of %d..%d, margin %.0f); "
            "best-of-runs -logL vs sgl: max excess %.2e (copies %.2e), max gain %.2e (copies %.2e), moved %d pairs "
            "(copies %d..%d); mean timesRun %.3f (copies %.3f..%.3f, reference %.3f); timesConverged > 1: %d pairs "
            "(copies %d..%d, reference %d)" % (
                name, st["on"], st["used"], *st["on_copies"], *st["used_copies"], st["margin"], st["excess"],
                st["excess_copies"], st["gain"], st["gain_copies"], st["moved"], *st["moved_copies"], st["mean_runs"],
                *st["mean_runs_copies"], st["mean_runs_ref"], st["multi"], *st["multi_copies"], st["multi_ref"]))
