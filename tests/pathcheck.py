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
