"""relate_hip_linkage (the reference's HMMtest on the GPU) through the C ABI: bit-for-bit equal to the CPU oracle on
the same inputs (float arithmetic in the reference's summation order), and equal to the reference program's own
printed outputs with its permutation seed pinned (tests/golden/hmmtest/golden.json)."""
import numpy as np
import pytest

import hmmtestcheck as hc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["default", "ccall2", "missing", "kfile", "mixed0", "mixed1", "mixed2"])
def test_linkage_equals_oracle_and_reference(gpu, name):
    G = hc.golden()
    case = G["cases"][name]
    inp = hc.case_inputs(case)
    want = hc.oracle_linkage(inp, G["seed"])
    got = gpu.linkage(inp["post"], inp["cc"], pos=inp["pos"], K=inp["K"], isK0=inp["isK0"], nsim=inp["nsim"],
                      ccAll=inp["ccAll"], sig=inp["sig"], infer=inp["infer"], seed=G["seed"], want_cov=True,
                      want_post=True)
    assert np.array_equal(got["post"], want["post"])       # fix_missing
    assert np.array_equal(got["cov"], want["cov"])         # covariance of posterior rows
    assert np.array_equal(got["stat"], want["stat"])       # observed statistic
    assert np.array_equal(got["xmax"], want["xmax"])       # permutation maxima
    for k in ("pmaxsim", "psim_over_nsim", "psigsim", "n_case_pairs", "n_control_pairs"):
        assert got[k] == want[k], k
    assert np.array_equal(hc.as_printed(got["stat"]), np.array(case["stat"]))
    assert np.array_equal(hc.as_printed(got["xmax"]), np.array(case["xmax"]))
    assert float("%f" % got["pmaxsim"]) == case["pmaxsim"]
    assert float("%f" % got["psim_over_nsim"]) == case["psim_over_nsim"]
    assert float("%f" % got["psigsim"]) == case["psigsim"]


def test_linkage_on_a_larger_synthetic_matrix(gpu):
    """40 individuals -> 780 pair rows x 3 000 SNPs with 2 % missing values, 300 permutations: the tiled covariance,
    the permutation batches and the missing-value paths at a size with ragged tiles; oracle time ~10 s."""
    rng = np.random.default_rng(11)
    n, M = 40, 3000
    R = n * (n - 1) // 2
    post = rng.uniform(0.0, 1.0, size=(R, M)).astype(np.float32)
    post[rng.random((R, M)) < 0.02] = -1
    post[3, :5] = -1
    post[7, -9:] = -1
    pos = np.cumsum(rng.uniform(0.01, 0.3, size=M)).astype(np.float32)
    cc = (rng.random(n) < 0.5).astype(np.int32)
    for infer in (1, 0):
        inp = dict(post=post, pos=pos if infer else None, K=None, isK0=0, cc=cc, nsim=300, ccAll=2, sig=0.05,
                   infer=infer)
        want = hc.oracle_linkage(inp, 99)
        got = gpu.linkage(post, cc, pos=inp["pos"], nsim=300, ccAll=2, infer=infer, seed=99, want_cov=True)
        assert np.array_equal(got["cov"], want["cov"], equal_nan=True)
        assert np.array_equal(got["stat"], want["stat"], equal_nan=True)
        assert np.array_equal(got["xmax"], want["xmax"], equal_nan=True)
        assert (got["pmaxsim"], got["psim_over_nsim"], got["psigsim"]) == (
            want["pmaxsim"], want["psim_over_nsim"], want["psigsim"])


def test_linkage_rejects_inconsistent_inputs(gpu):
    import relate_amd
    post = np.zeros((10, 8), dtype=np.float32)
    with pytest.raises(relate_amd.RelateError):
        gpu.linkage(post, np.array([0, 1, 0, 1], dtype=np.int32), infer=0)       # 4 individuals = 6 pairs, not 10
    with pytest.raises(relate_amd.RelateError):
        gpu.linkage(post, np.array([0, 1, 0, 1, 1], dtype=np.int32), infer=1)    # infer needs positions


def test_decode_to_linkage_on_the_device_equals_the_route_through_the_post_text(gpu):
    """relate_hip_linkage_pairs (decode + HMMtest, posterior rows never leave HBM) against the route the reference's
    pipeline takes on the same estimates: decode to the host, the .post text (write_dMatrix_to_file's %.4g, through
    relate_format_post_rows), the text read back as floats, relate_hip_linkage.  Config 5: dat_small, options.linkage.txt,
    the 20 keep-list individuals and their case/control file.  digits = 4 rounds the double posteriors on the device as
    the text does, so the posterior matrix after inference, the statistic and the permutation maxima are the file
    route's bit for bit; digits = 0 differs by the rounding of the text, < 1e-3."""
    import os
    import pathcheck as pc
    import refload
    import relate_amd
    g, pos, chr_ = pc.dataset("dat_small")
    opt = dict(pc.options("options.linkage.txt"), seed=4242, seed_mode=0)
    M = gpu.init_globals(g, pos, chr_, opt)
    pairs = [tuple(int(v) for v in p) for p in refload.keep_pairs(os.path.join(refload.DATA, "dat_small.keep"))]
    cc = np.array([int(v) for v in open(os.path.join(refload.DATA, "dat_small.cc")).read().split()], dtype=np.int32)
    assert len(pairs) == 190 and len(cc) == 20
    res = gpu.fit(pairs)
    par = np.array([(r["a"], r["k2"], r["k1"]) for r in res])
    _, spos, _ = gpu.stripped()
    fpos = spos.astype(np.float32)
    # file route
    post, _, _ = gpu.decode(pairs, par)
    import io as _io
    from relate_amd import io as rio
    buf = _io.BytesIO()
    rio.write_post_block(buf, post, True)       # one row per pair (fixK2 = 1, value 0): the IBD0 posteriors
    text = np.array([[float(v) for v in r.split(b"\t") if v.strip()] for r in buf.getvalue().splitlines()],
                    dtype=np.float32)
    assert text.shape == (190, M)
    kw = dict(nsim=200, ccAll=0, infer=1, seed=7)
    want = gpu.linkage(text, cc, pos=fpos, want_post=True, **kw)
    # device route
    got = gpu.linkage_pairs(pairs, par, cc, digits=4, pos=fpos, want_post=True, **kw)
    assert np.array_equal(got["post"], want["post"])
    assert np.array_equal(got["stat"], want["stat"]) and np.array_equal(got["xmax"], want["xmax"])
    assert (got["n_case_pairs"], got["n_control_pairs"]) == (want["n_case_pairs"], want["n_control_pairs"])
    assert (got["pmaxsim"], got["psim_over_nsim"], got["psigsim"]) == (want["pmaxsim"], want["psim_over_nsim"],
                                                                      want["psigsim"])
    raw = gpu.linkage_pairs(pairs, par, cc, digits=0, want_post=True, **kw)     # positions from the handle
    np.testing.assert_allclose(raw["stat"], want["stat"], rtol=2e-2, atol=2e-3)
    ok = want["post"] != 0
    np.testing.assert_allclose(raw["post"][ok], want["post"][ok], rtol=1e-3, atol=1e-7)
