"""The linkage oracle (oracle/hmmtest_oracle.c) against outputs of the reference's own HMMtest program with its
permutation seed pinned (tests/golden/hmmtest/golden.json): the statistic, the permutation maxima and the first
rows of the covariance matrix agree with the reference's printed 6 digits exactly, the three summary values too."""
import numpy as np
import pytest

import hmmtestcheck as hc


@pytest.mark.parametrize("name", ["default", "ccall2", "missing", "kfile", "mixed0", "mixed1", "mixed2"])
def test_oracle_reproduces_reference_output(name):
    G = hc.golden()
    case = G["cases"][name]
    got = hc.oracle_linkage(hc.case_inputs(case), G["seed"])
    assert got["n_case_pairs"] == case["n_case_pairs"] and got["n_control_pairs"] == case["n_control_pairs"]
    assert np.array_equal(hc.as_printed(got["stat"]), np.array(case["stat"]))
    assert np.array_equal(hc.as_printed(got["xmax"]), np.array(case["xmax"]))
    cov = np.array(case["cov_first_rows"])
    assert np.array_equal(hc.as_printed(got["cov"][:len(cov)].ravel()), cov.ravel())
    assert float("%f" % got["pmaxsim"]) == case["pmaxsim"]
    assert float("%f" % got["psim_over_nsim"]) == case["psim_over_nsim"]
    assert float("%f" % got["psigsim"]) == case["psigsim"]
