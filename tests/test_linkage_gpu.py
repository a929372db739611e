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
