"""relate_amd/csrc/dd_math.hpp (the correctly rounded acos / cos / pow of the LD cubic solve) compiled for the host
and checked against 200-bit arithmetic (mpmath): every result must be the correctly rounded double.  The same header
is what the HIP kernel compiles; tests/test_gpu_parity.py checks the resulting LD tables on the device."""
import ctypes as C
import math
import os
import random
import subprocess

import pytest

mp = pytest.importorskip("mpmath")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("ddm")
    src = d / "ddm.cpp"
    src.write_text('#include "%s"\n'
                   'extern "C" {\n'
                   'double dd_cos(double x) { return ddm::cos_cr(x); }\n'
                   'double dd_acos(double x) { return ddm::acos_cr(x); }\n'
                   'double dd_pow(double x, double y) { return ddm::pow_cr(x, y); }\n'
                   'double dd_log(double x) { return ddm::log_cr(x); }\n}\n'
                   % os.path.join(ROOT, "relate_amd", "csrc", "dd_math.hpp"))
    so = d / "libddm.so"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so), str(src)])
    L = C.CDLL(str(so))
    for f in (L.dd_cos, L.dd_acos, L.dd_log):
        f.restype, f.argtypes = C.c_double, [C.c_double]
    L.dd_pow.restype, L.dd_pow.argtypes = C.c_double, [C.c_double, C.c_double]
    return L


def test_correctly_rounded(lib):
    mp.mp.prec = 200
    rng = random.Random(11)
    third = 1.0 / 3.0
    hard_cos = [math.pi / 2, -math.pi / 2, math.pi, -math.pi, math.pi / 3, 2 * math.pi / 3, 1e-9, 0.0, 4.0, -4.0]
    agree = [0, 0, 0, 0]
    n = 12000
    for i in range(n):
        x = rng.uniform(-math.pi, math.pi) if i % 5 else rng.choice(hard_cos) * (1 + rng.choice([0, 1, -1, 3]) * 1.1e-16)
        assert lib.dd_cos(x) == float(mp.cos(mp.mpf(x))), x
        agree[0] += lib.dd_cos(x) == math.cos(x)
        y = rng.uniform(-1, 1) if i % 5 else rng.choice([1 - 10 ** rng.uniform(-16, -1), -1 + 10 ** rng.uniform(-16, -1),
                                                         0.7, -0.7, 0.0, 1e-300, 0.7000000000000001])
        assert lib.dd_acos(y) == float(mp.acos(mp.mpf(y))), y
        agree[1] += lib.dd_acos(y) == math.acos(y)
        z = 10 ** rng.uniform(-30, 5)
        assert lib.dd_pow(z, third) == float(mp.power(mp.mpf(z), mp.mpf(third))), z
        agree[2] += lib.dd_pow(z, third) == math.pow(z, third)
        w = rng.uniform(0, 1) if i % 3 else 1 + rng.choice([-1, 1]) * 10 ** rng.uniform(-16, -3)
        w = w if w > 0 else 0.5
        assert lib.dd_log(w) == float(mp.log(mp.mpf(w))), w
        agree[3] += lib.dd_log(w) == math.log(w)
    # the host libm (what the reference links) is itself correctly rounded for all but ~0.1 % of arguments
    assert min(agree) > 0.995 * n
    assert lib.dd_acos(1.0) == 0.0 and lib.dd_acos(-1.0) == math.pi and math.isnan(lib.dd_acos(1.5))
