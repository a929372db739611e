"""The product's re-entrant L-BFGS-B + numeric-gradient driver (relate_findmax_bfgs, relate_amd/csrc/lbfgsb.hpp,
fit.hpp) against the reference's findmax_bfgs (src/bfgs.c:127-199): golden results everywhere, the compiled
reference where available.  Host code only -- no GPU.  Results must be bit-identical (the optimiser's decisions
are only reproducible if every iterate is)."""
import numpy as np
import pytest

import pathcheck as pc
import relate_amd

FUNS = {
    "f1": lambda x: (x[0] - 0.3) ** 2 * 7 + (x[1] - 0.8) ** 4 + 0.5 * x[0] * x[1] + 3,
    "rosen": lambda x: 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2 + (x[2] - 0.2) ** 2,
    "f3": lambda x: float(np.exp(x[0]) - 2 * x[0]),
}


def test_rand_stream_is_glibc(ref=None):
    g = pc.golden()["rand_stream"]
    got = relate_amd.rand_stream(g["seed"], len(g["values"]))
    assert [int(v) for v in got] == g["values"]
    assert [int(v) for v in relate_amd.rand_stream(1, 3)] == [1804289383, 846930886, 1681692777]


@pytest.mark.parametrize("case", range(len(pc.golden()["findmax"])))
def test_findmax_golden(case):
    c = pc.golden()["findmax"][case]
    x, r = relate_amd.findmax_bfgs(FUNS[c["fun"]], c["x0"], c["lo"], c["hi"])
    assert [float(v) for v in x] == c["x"]
    assert r == c["ret"]


def test_findmax_vs_reference_more_functions(ref):
    rng = np.random.default_rng(2)
    cases = []
    for _ in range(12):
        c = rng.uniform(-0.5, 1.5, size=3)
        w = rng.uniform(0.5, 20, size=3)
        q = rng.uniform(-1, 1)

        def f(x, c=c, w=w, q=q):
            return float(sum(w[i] * (x[i] - c[i]) ** 2 for i in range(len(x))) + q * x[0] * x[-1] + np.cos(3 * x[0]))
        for n in (1, 2, 3):
            cases.append((f, list(rng.uniform(0, 1, size=n)), [0.0] * n, [1.0] * n))
    # discontinuous objective like calcA's infeasible region (returns 1e7)
    def g(x):
        return 1e7 if (x[1] + 2 * x[0]) ** 2 < 4 * x[0] else float((x[0] - 0.1) ** 2 + (x[1] - 0.6) ** 2 + 5)
    cases.append((g, [0.05, 0.7], [0, 0], [1, 1]))
    cases.append((g, [0.02, 0.3], [0, 0], [1, 1]))
    for f, x0, lo, hi in cases:
        xr, fr = ref.findmax_bfgs(f, x0, lo, hi)
        xm, fm = relate_amd.findmax_bfgs(f, x0, lo, hi)
        assert np.array_equal(xr, xm), (x0, xr, xm)
        assert fr == fm
