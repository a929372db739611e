"""The HIP path against the compiled reference run live (oracle/_ref: init_globals + sgl on the GPU box's host cores)
over synthetic data sets and option variations that the golden fixtures do not cover.  Per configuration: SNPs kept
after MAF filter / pruning, the pair's kept SNPs and distances, emissions, the three likelihoods sgl reports
(unrelated, parent-offspring, at the doPar point), posteriors and Viterbi path.  Then the multi-start fit in four
optimisation modes with a pinned seed.  `pytest -s` prints one line per case (tools/twin_sweep.py does that and keeps
the log under profiles/).

Left out on purpose: back = 1 (remove_cross zeroes a row at matrix depth 2 instead of shifting: SURVEY 8a quirk 12,
excluded from parity) and ld_adj = 0 (the reference's sgl segfaults on it in this harness; covered against the oracle
in test_gpu_parity.py)."""
import numpy as np
import pytest

import pathcheck as pc

pytestmark = pytest.mark.gpu

F32 = lambda v: float(np.float32(v))  # noqa: E731  (option values pass through float in the reference)
CASES = [
    ("N12 M2000", dict(N=12, M=2000), {}),
    ("N50 M2500 miss 3%", dict(N=50, M=2500, miss=0.03), {}),
    ("N24 M3000 back 2", dict(N=24, M=3000), dict(back=2, back2=4)),
    ("N24 M3000 back 12", dict(N=24, M=3000, miss=0.01), dict(back=12, back2=24)),
    ("N24 M3000 back 20 (generic staging)", dict(N=24, M=3000, miss=0.01), dict(back=20, back2=40)),
    ("N30 M3000 LD=0", dict(N=30, M=3000), dict(LD=0)),
    ("N30 M3000 double_recom", dict(N=30, M=3000), dict(double_recom=1)),
    ("N30 M3000 eps 0.05 min 0.1", dict(N=30, M=3000), dict(epsilon=F32(0.05), min=F32(0.1))),
    ("N40 M4000 prune", dict(N=40, M=4000), dict(doPrune=1, prune_val=F32(0.1), back=10, back2=20)),
    ("N16 M1500 copy 0.6", dict(N=16, M=1500, copy_p=0.6), {}),
    ("N3 M1200", dict(N=3, M=1200), {}),
    ("N5 M1500 min 0.2", dict(N=5, M=1500), dict(min=F32(0.2))),
    ("N20 M2000 miss 8%", dict(N=20, M=2000, miss=0.08), {}),
    ("N20 M1800 one chromosome", dict(N=20, M=1800, nchr=1), {}),
    ("N300 M600", dict(N=300, M=600), {}),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_stage_like_decode_at_a_point(gpu, ref, case):
    from relate_amd import synth
    name, dk, ok = case
    kw = dict(dk)
    N, M = kw.pop("N"), kw.pop("M")
    g, pos, chr_, related = synth.make_dataset(N, M, n_related=min(3, N // 2), seed=20240601 + len(name), **kw)
    opt = dict(pc.options("options.single.pair.txt"), **ok)
    opt.update(doPar=1, par=[0.05, 0.1, 0.3])   # no optimisation: likelihood, posteriors and path at (a, k2, k1)
    Mref = ref.init(g.astype(np.int32), pos, chr_, opt)
    try:
        gpu.init_globals(g, pos, chr_, dict(opt, seed=1, seed_mode=0))
        assert gpu.M == Mref
        hg, hr = gpu.hap_global(), ref.hap_global()
        neq = max(float((hg[k] != hr[k]).mean()) for k in ("mea", "pBA", "pBa", "pbA", "pba"))
        assert neq < 0.005                      # LD cells differing in the last bits (dd_math.hpp)
        pairs = [(int(r[0]), int(r[1])) for r in related] + [(0, N - 1), (1, N // 2)]
        pairs = sorted({p for p in pairs if p[0] < p[1]})
        res, post, path = gpu.sgl(pairs, want_path=True)
        worst = lk = pw = 0.0
        for n, (i, j) in enumerate(pairs):
            x = ref.sgl(i, j, seed=1)
            kp, tt, S = gpu.pair_emissions(i, j)
            idx = np.where(kp == 1)[0]
            assert np.array_equal(kp, x["keep"])
            assert np.array_equal(tt[idx][1:], x["t"][1:x["m"]])
            worst = max(worst, float(np.abs(S[:, idx] / x["S"] - 1).max()))
            lk = max(lk, abs(res[n]["uLike"] / x["uLike"] - 1), abs(res[n]["poLike"] / x["poLike"] - 1),
                     abs(res[n]["negloglik"] / x["nll"] - 1))
            pw = max(pw, float(np.abs(post[n][:, idx] - x["post"]).max()))
            assert (post[n][:, kp == 0] == -1).all()
            assert np.array_equal(path[n][idx], x["path"])
        print("\n%-38s M %d  S maxrel %.2e  like (u, po, par) maxrel %.2e  post maxabs %.1e  LD cells differing %.3f %%" % (
            name, gpu.M, worst, lk, pw, 100 * neq))
        assert worst < 1e-12 and lk < 1e-12 and pw < 1e-9
    finally:
        ref.fini()


# (name, option overrides, pairs that must stay on the reference's trajectory, largest relative difference of the -logL
#  reached, relative -logL agreement that counts as "on the trajectory", relative tolerance of the estimates on it)
MODES = [
    ("3 free (a, k2, k1)", dict(calcA=0, fixA=0, fixK=0), 6, 1e-4, 1e-9, 1e-6),
    ("fixK2=0: a, k1 free", dict(calcA=0, fixA=0, fixK=1, fixK_val=0.0), 6, 1e-4, 1e-9, 1e-6),
    ("fixK2=0 + calcA: k1 free", dict(calcA=1, fixA=0, fixK=1, fixK_val=0.0), 4, 1e-4, 1e-9, 1e-6),
    ("fixK2=0.1: a, k1 free", dict(calcA=0, fixA=0, fixK=1, fixK_val=F32(0.1)), 0, 1e-4, 1e-9, 1e-6),
    # fixK2_fixA_fun / fixK2_fixA_optim (src/relateHMM.cpp:1783-1825): k1 alone is free, a and k2 are given.  One
    # dimension and a smooth optimum: the search stops when the projected gradient is below 1e-4 (PGTOL, src/bfgs.h:46),
    # which leaves k1 free in its 6th digit while -logL agrees to 1e-12 (measured: k1 1.2e-6 apart, relative), so the
    # estimates of a pair on the trajectory are held to 1e-5 here.  With k2 = 0.1 the optimum of unrelated pairs lies on the infeasibility wall
    # (k1 + 2 k2)^2 = 4 k2 as for fixK2 = 0.1 above
    ("fixK2=0 + fixA=0.05: k1 free", dict(calcA=0, fixA=1, fixA_val=F32(0.05), fixK=1, fixK_val=0.0, timesToRun=8), 4, 1e-4, 1e-12, 1e-5),
    ("fixK2=0.1 + fixA=0.02: k1 free", dict(calcA=0, fixA=1, fixA_val=F32(0.02), fixK=1, fixK_val=F32(0.1), timesToRun=8), 0, 1e-3, 1e-12, 1e-5),
    # calcA_fun / calcA_optim (:1564-1621): k2 and k1 free, a = calculateA -- the mode of configs 2, 3 and 4, chaotic:
    # the bar on -logL is the largest excess the reference shows against itself under objective noise of the HIP
    # objective's size (tests/golden/sensitivity_allpair.json: 4.2e-3); per-run bars in tests/test_gpu_sensitivity.py
    ("calcA: k2, k1 free", dict(calcA=1, fixA=0, fixK=0, timesToRun=8, alim1=1.0), 0, 4.2e-3, 1e-9, 1e-6),
]


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
def test_fit_modes(gpu, ref, mode):
    """Pinned seed, same start points.  In every mode the objective at the reference's own estimate agrees to 1e-12
    and pairs on the reference's trajectory (-logL within 1e-9) have its estimates and posteriors.  How many pairs
    stay on it depends on the mode: all with 3 free parameters or k2 fixed at 0; the planted related pairs under
    calcA (unrelated pairs sit on a flat stretch at k1 ~ 7e-4); with k2 fixed at 0.1 the optimum lies on the
    infeasibility wall (k1 + 2 k2)^2 = 4 k2, where the reference's runs do not converge either (8 of 8 restarts used),
    and the -logL reached differs in the 5th digit in both directions."""
    from relate_amd import synth
    name, ok, min_on, max_rel, on_rel, est_rel = mode
    N, M, seed = 24, 3000, 777
    g, pos, chr_, related = synth.make_dataset(N, M, n_related=4, miss=0.005)
    pairs = [(int(r[0]), int(r[1])) for r in related] + [(0, N - 1), (1, N // 2)]
    opt = dict(pc.options("options.single.pair.txt"), **ok)
    ref.init(g.astype(np.int32), pos, chr_, opt)
    try:
        gpu.init_globals(g, pos, chr_, dict(opt, seed=seed, seed_mode=0))
        res, post, _ = gpu.sgl(pairs)
        on, worst, obj = 0, 0.0, 0.0
        for n, (i, j) in enumerate(pairs):
            x, r = ref.sgl(i, j, seed=seed), res[n]
            at_ref = gpu.like([(i, j)], np.array([[[x["a"], x["k2"], x["k1"]]]]))[0, 0]
            obj = max(obj, abs(at_ref / x["nll"] - 1))
            rel = abs(r["negloglik"] / x["nll"] - 1)
            worst = max(worst, rel)
            assert rel < max_rel
            if rel <= on_rel:
                on += 1
                for k in ("a", "k2", "k1"):
                    assert r[k] == pytest.approx(x[k], rel=est_rel, abs=1e-7), k
                idx = np.where(x["keep"] == 1)[0]
                np.testing.assert_allclose(post[n][:, idx], x["post"], atol=1e-6)
        print("\nfit %-30s objective at the reference's estimate maxrel %.1e; -logL reached maxrel %.2e; on the reference's "
              "trajectory %d/%d" % (name, obj, worst, on, len(pairs)))
        assert obj < 1e-12
        assert on >= min_on
    finally:
        ref.fini()
