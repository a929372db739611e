"""Pins the CPU oracle (oracle/relate_oracle.c) against the golden fixtures generated from the compiled
reference (tests/golden/make_golden.py).  Runs everywhere (no GPU, no reference library needed).

Tolerances: integer/index results exact; like() and posteriors exact to 1e-13 relative / absolute (they are
bit-identical here, the slack covers a different libm on another box); LD-conditioned emissions 1e-13 relative
(the oracle sums the 81 terms by their generating rule, the reference term by term)."""
import numpy as np
import pytest

import pathcheck as pc

BLOCKS = list(range(4))


@pytest.fixture(scope="module")
def prepared(oracle):
    out = {}
    for b in pc.golden()["blocks"]:
        key = (b["data"], b["options"])
        g, pos, chr_ = pc.dataset(b["data"])
        opt = pc.options(b["options"])
        out[key] = (opt, pc.oracle_precalc(oracle, g, pos, opt))
    return out


def test_ld_polynomials_match_reference_coefficients(oracle):
    """All 81 x 3 polynomials of LDemission_new (src/relateHMM.cpp:2343-2704): the oracle's generating rule,
    evaluated with a unit weight on one true-genotype combination, equals the reference's coefficients."""
    coef = pc.poly_golden()
    rng = np.random.default_rng(3)
    # weights err(tr|obs) with eps = 0 pick exactly one true genotype per individual and SNP:
    # obs code o -> true index o-1 gets (1-eps)^2 = 1, except the reference's asymmetric het row (2*(1-e)*e = 0)
    for _ in range(5):
        A, w, v, x = rng.uniform(0.05, 0.6, size=4)
        for c in range(81):
            i, j, k, l = c // 27, (c // 9) % 3, (c // 3) % 3, c % 3
            got = oracle.emis_ld(A, w, v, x, i + 1, j + 1, k + 1, l + 1, 0.0)
            for row in range(3):
                want = sum(kk * np.prod(np.array([A, w, v, x]) ** np.array(e)) for kk, e in coef["%d,%d" % (c, row)])
                assert got[row] == pytest.approx(want, rel=1e-13, abs=1e-300), (c, row)


def test_error_table_is_the_references(oracle):
    e = 0.01
    want = {(0, 1): (1 - e) ** 2, (0, 2): (1 - e) * e, (0, 3): e * e, (1, 1): 2 * (1 - e) * e,
            (1, 2): (1 - e) ** 2 + e * e, (1, 3): 2 * (1 - e) * e, (2, 1): e * e, (2, 2): (1 - e) * e,
            (2, 3): (1 - e) ** 2}
    for (tr, o), val in want.items():
        assert oracle.L.orc_err(tr, o, e) == pytest.approx(val, rel=1e-15)


@pytest.mark.parametrize("bi", BLOCKS)
def test_precalc(oracle, prepared, bi):
    b = pc.golden()["blocks"][bi]
    opt, p = prepared[(b["data"], b["options"])]
    assert int(p["keep"].sum()) == b["M"]
    assert [int(i) for i in np.where(p["keep"] == 1)[0]] == b["keep_idx"]
    assert np.array_equal(p["mafs"][::37], np.array(b["maf_sub"]))
    for d, s, mea, pBA, pBa, pbA, pba in b.get("hap_cells", []):
        h = p["hap"]
        got = (h["mea"][d, s - 1], h["pBA"][d, s], h["pBa"][d, s], h["pbA"][d, s], h["pba"][d, s])
        assert np.allclose(got, (mea, pBA, pBa, pbA, pba), rtol=1e-13, atol=1e-15, equal_nan=True), (d, s)


@pytest.mark.parametrize("bi", BLOCKS)
def test_stage_like_decode_viterbi(oracle, prepared, bi):
    b = pc.golden()["blocks"][bi]
    opt, p = prepared[(b["data"], b["options"])]
    grid = pc.golden()["grid"]
    for rec in b["pairs"]:
        i, j = rec["pair"]
        st = pc.oracle_stage(oracle, p, opt, i, j)
        assert st["m"] == rec["m"]
        assert [int(s) for s in np.where(st["keep"] == 0)[0]] == rec["dropped_idx"]
        sub = rec["sub"]
        assert np.array_equal(st["t"][sub], np.array(rec["t_sub"]))
        assert [int(st["choose"][s]) for s in sub] == rec["choose_sub"]
        for r in range(3):
            assert np.array_equal(st["S1"][r][sub], np.array(rec["S1_sub"][r]))
            np.testing.assert_allclose(st["S"][r][sub], np.array(rec["S_sub"][r]), rtol=1e-13)
        Sk = st["Sk"]
        got = np.array([oracle.like(d, *Sk, st["t"], opt["double_recom"], opt["alim1"]) for d in grid])
        np.testing.assert_allclose(got, np.array(rec["like_grid"]), rtol=1e-12)
        for dec in rec["decode"]:
            post = oracle.decode(dec["par"], *Sk, st["t"], opt["double_recom"])
            for r in range(3):
                np.testing.assert_allclose(post[r][sub], np.array(dec["post_sub"][r]), rtol=0, atol=1e-9)
            path = oracle.viterbi(dec["par"], *Sk, st["t"], opt["double_recom"])
            assert "".join(str(int(v)) for v in path) == dec["path"]


def test_like_rejections(oracle):
    Sk = [np.log(np.full(5, 0.1))] * 3
    t = np.array([0, 1, 1, 1, 1, 0.0])
    for d in ((0.1, -0.01, 0.2), (0.1, 0.2, -0.1), (0.1, 0.6, 0.6), (0.1, 1.5, 0.0), (0.2, 0.1, 0.1)):
        assert oracle.like(d, *Sk, t, 0, 0.15) == 10000000.0


def test_calculate_a(oracle):
    # calculateA (src/relateHMM.cpp:974-1007): k2 == 0 branch and the general branch
    phi = float(np.float32(0.013))
    assert oracle.calculateA(0.5, 0.5, 0.0, phi) == pytest.approx(-(1 - np.log(0.5) / np.log(2)) * np.log(1 - phi))
    k2, k1 = 0.25, 0.5
    xa = (k1 + 2 * k2 + np.sqrt((k1 + 2 * k2) ** 2 - 4 * k2)) / 2
    xb = k2 / xa
    m = (1 - np.log(xa) / np.log(2)) + (1 - np.log(xb) / np.log(2))
    assert oracle.calculateA(0.25, k1, k2, phi) == pytest.approx(-m * np.log(1 - phi), rel=1e-14)
