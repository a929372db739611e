"""Oracle versus the compiled reference itself (oracle/_ref, built from /root/reference/src by oracle/Makefile).
Skipped where the library is not available.  Wider coverage than the golden fixtures: every pair of the keep list,
random LD cells including every root-finding branch of do_geno_cal, remove_cross on pairs with missing data."""
import numpy as np
import pytest

import pathcheck as pc


def _cells_for_branches():
    """3x3 count tables hitting each branch of do_geno_cal (src/ld.cpp:356-437): e == 0; jm == kl == 0;
    jm == 0; kl == 0; cubic with three real roots; cubic with one real root; double-root window."""
    tabs = [
        [10, 3, 2, 4, 0, 5, 1, 2, 9],      # e == 0
        [0, 0, 0, 0, 7, 0, 0, 0, 0],       # only double hets: jm == kl == 0
        [0, 2, 5, 3, 6, 0, 4, 0, 0],       # j*m == 0
        [6, 1, 0, 2, 5, 3, 0, 1, 7],       # k*l == 0 ? (c = g = 0 is not enough; keeps b,d,f,h) -> generic
        [9, 0, 0, 0, 4, 0, 0, 0, 8],       # kl == 0, jm != 0
        [12, 5, 1, 6, 9, 4, 2, 3, 11],     # generic cubic
        [1, 8, 14, 7, 10, 6, 13, 9, 2],    # generic cubic, negative D
        [30, 1, 0, 1, 1, 1, 0, 1, 30],     # strong LD
        [3, 3, 3, 3, 3, 3, 3, 3, 3],       # flat
    ]
    cols = []
    for t in tabs:
        a, b = [], []
        for idx, n in enumerate(t):
            a += [idx // 3 + 1] * n
            b += [idx % 3 + 1] * n
        a += [0, 1, 0]
        b += [2, 0, 0]   # missing genotypes are skipped
        cols.append((np.array(a), np.array(b)))
    return cols


def test_ld_cell_branches(oracle, ref):
    for a, b in _cells_for_branches():
        want = ref.ld_cell(a, b)
        got = oracle.ld_cell(a, b)
        np.testing.assert_allclose(got, want, rtol=1e-14, atol=0, equal_nan=True)


def test_ld_cell_random(oracle, ref):
    rng = np.random.default_rng(11)
    for _ in range(300):
        n = int(rng.integers(5, 120))
        p = rng.uniform(0.05, 0.95)
        a = rng.binomial(2, p, size=n) + 1
        b = np.where(rng.random(n) < 0.6, a, rng.binomial(2, rng.uniform(0.05, 0.95), size=n) + 1)
        a[rng.random(n) < 0.03] = 0
        b[rng.random(n) < 0.03] = 0
        want = ref.ld_cell(a, b)
        got = oracle.ld_cell(a, b)
        np.testing.assert_allclose(got, want, rtol=1e-14, atol=0, equal_nan=True)


@pytest.mark.parametrize("optf", ["options.single.pair.txt", "options.linkage.txt"])
def test_whole_stage_all_keep_pairs(oracle, ref, optf):
    g, pos, chr_ = pc.dataset("dat_small")
    opt = pc.options(optf)
    p = pc.oracle_precalc(oracle, g, pos, opt)
    M = ref.init(g, pos, chr_, dict(opt, doPar=1))   # doPar: no optimisation, likelihood/decode at opt["par"]
    try:
        keep, genos, maf, posg = ref.globals()
        assert M == p["gs"].shape[1]
        assert np.array_equal(keep, p["keep"])
        assert np.array_equal(genos.astype(np.uint8), p["gs"])
        assert np.array_equal(maf, p["mafs"])
        assert np.array_equal(posg, p["ps"])
        hr = ref.hap_global()
        for k in ("mea", "pBA", "pBa", "pbA", "pba"):
            assert np.array_equal(hr[k], p["hap"][k], equal_nan=True), k
        import refload
        pairs = refload.keep_pairs(refload.DATA + "/dat_small.keep")[::7] + [(100, 103), (20, 99)]
        for (i, j) in pairs:
            x = ref.sgl(i, j)
            st = pc.oracle_stage(oracle, p, opt, i, j)
            assert st["m"] == x["m"]
            assert np.array_equal(st["keep"], x["keep"])
            assert np.array_equal(st["t"], x["t"])
            assert np.array_equal(st["choose"], x["choose"])
            assert np.array_equal(st["S1"], x["S1"])
            np.testing.assert_allclose(st["S"], x["S"], rtol=1e-13)
            par = list(opt["par"])
            nll = oracle.like(par, *st["Sk"], st["t"], 0, opt["alim1"])
            assert nll == pytest.approx(x["nll"], rel=1e-12)
            post = oracle.decode(par, *st["Sk"], st["t"], 0)
            np.testing.assert_allclose(post, x["post"], rtol=0, atol=1e-9)
            path = oracle.viterbi(par, *st["Sk"], st["t"], 0)
            assert np.array_equal(path, x["path"])
    finally:
        ref.fini()


def test_double_recombination_kernels(oracle, ref):
    g, pos, chr_ = pc.dataset("dat_small")
    opt = pc.options("options.single.pair.txt")
    p = pc.oracle_precalc(oracle, g, pos, opt)
    st = pc.oracle_stage(oracle, p, opt, 5, 12)
    for par in ((0.05, 0.2, 0.3), (0.1, 0.0, 0.4)):
        a = ref.like(par, *st["Sk"], st["t"], 1, [opt["alim0"], opt["alim1"]])
        b = oracle.like(par, *st["Sk"], st["t"], 1, opt["alim1"])
        assert b == pytest.approx(a, rel=1e-13)
        np.testing.assert_allclose(oracle.decode(par, *st["Sk"], st["t"], 1), ref.decode(par, *st["Sk"], st["t"], 1),
                                   atol=1e-10)
        assert np.array_equal(oracle.viterbi(par, *st["Sk"], st["t"], 1), ref.viterbi(par, *st["Sk"], st["t"], 1))


def test_pruning_mask(oracle, ref):
    g, pos, chr_ = pc.dataset("dat_small")
    opt = pc.options("options.linkage.txt")
    g8 = g.astype(np.uint8)
    keep = oracle.maf_keep(oracle.maf(g8), opt["min"])
    gs = np.ascontiguousarray(g8[:, keep == 1])
    for back, pv in ((50, opt["prune_val"]), (5, 0.05), (3, 0.2), (2, 0.1)):
        want = ref.prune(gs.astype(np.int32), opt["LD"], back, pv)
        got = oracle.prune(gs, opt["LD"], back, pv)
        assert np.array_equal(got, want), (back, pv)
