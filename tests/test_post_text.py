"""relate_format_post_rows (the .post text writer, write_dMatrix_to_file src/relateHMM.cpp:117-136) against printf's
"%.4g" -- Python's % operator formats with correctly rounded decimal conversion, as glibc's printf and the reference's
ostream do.  Host code only: runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

import relate_amd


def fmt_rows(post, k1_only=False):
    L = relate_amd.load_library()
    L.relate_post_text_stride.restype = C.c_size_t
    L.relate_post_text_stride.argtypes = [C.c_int]
    L.relate_format_post_rows.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_char_p,
                                          C.POINTER(C.c_size_t)]
    post = np.ascontiguousarray(post, dtype=np.float64)
    P, _, M = post.shape
    stride = L.relate_post_text_stride(M)
    rows = P * (1 if k1_only else 2)
    out = C.create_string_buffer(rows * stride)
    lens = (C.c_size_t * rows)()
    assert L.relate_format_post_rows(post.ctypes.data_as(C.POINTER(C.c_double)), P, M, int(k1_only), out, lens) == 0
    raw = out.raw
    return [raw[r * stride:r * stride + lens[r]].decode() for r in range(rows)]


def want_row(row):
    return "".join(("-1" if v == -1.0 else "%.4g" % v) + "\t" for v in row) + "\n"


def adversarial():
    v = [0.0, 1.0, 0.5, 0.25, 0.125, 0.0625, 0.03125, 0.28125, 0.71875, 9.9995e-5, 9.99949999e-5, 9.9996e-5, 1e-4,
         1.00005e-4, 0.99995, 0.999949999, 0.99996, 0.12345, 0.12335, 0.12355, 1e-5, 1.2345e-5, 9.9995e-6, 1e-290,
         1e-291, 5e-324, 2.2250738585072014e-308, 1e-100, 1.2355e-100, 9.9995e-101, 1e22, 1e23, 9999.5, 9999.4999,
         1234.5, 1235.5, 999.95, 99.995, 9.9995, 1000.0, 100.0, 10.0, 1e4, 12345.0, 1e5, 1e289, 1e290, 1e300, np.inf,
         np.nan, -0.0, -0.5, -1.0, -2.0, 0.1, 0.2, 0.3, 1 / 3, 2 / 3, 0.001, 0.0001, 0.00012344999999, 1e-4 * (1 - 1e-16)]
    v += [k / 8192 for k in range(0, 8193, 7)]                       # dyadic rationals: exact decimal ties
    v += [(2 * k + 1) * 5e-5 for k in range(0, 2000, 3)]             # decimal "ties" that are not exact in binary
    v += [10.0 ** k for k in range(-300, 300, 7)]
    v += [np.nextafter(10.0 ** k, 0) for k in range(-30, 30)] + [np.nextafter(10.0 ** k, np.inf) for k in range(-30, 30)]
    return np.array(v, dtype=np.float64)


def test_special_values_and_ties():
    v = adversarial()
    M = len(v)
    post = np.zeros((1, 3, M))
    post[0, 2] = v
    post[0, 1] = v[::-1]
    rows = fmt_rows(post)
    assert rows[0] == want_row(post[0, 2])
    assert rows[1] == want_row(post[0, 1])
    assert fmt_rows(post, k1_only=True) == [want_row(post[0, 2])]


@pytest.mark.parametrize("seed", [1, 2])
def test_random_values_equal_printf(seed):
    rng = np.random.default_rng(seed)
    P, M = 12, 70001
    post = np.empty((P, 3, M))
    post[:, 0] = rng.random((P, M))                                           # never printed
    post[:, 1] = 10.0 ** rng.uniform(-320, 5, (P, M))                          # all exponents, denormals included
    post[:, 2] = rng.random((P, M))                                           # what posteriors look like
    post[::3, 2] = np.round(rng.random((len(post[::3]), M)), 4)               # values that are 4-digit decimals already
    post[1::4, 1] = np.round(rng.random((len(post[1::4]), M)), 5)             # 5-digit decimals: near-ties everywhere
    post[:, 1:][rng.random((P, 2, M)) < 0.01] = -1.0                          # dropped SNPs
    rows = fmt_rows(post)
    assert len(rows) == 2 * P
    for p in range(P):
        assert rows[2 * p] == want_row(post[p, 2]), p
        assert rows[2 * p + 1] == want_row(post[p, 1]), p


def test_bad_arguments_are_rejected():
    L = relate_amd.load_library()
    L.relate_format_post_rows.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_char_p,
                                          C.POINTER(C.c_size_t)]
    assert L.relate_format_post_rows(None, 1, 1, 0, None, None) < 0


@pytest.mark.parametrize("k1_only", [False, True])
def test_block_writer_equals_row_writer(tmp_path, k1_only):
    """relate_amd.io.write_post_block (library, binary-mode file) writes the bytes of write_post_rows (pure Python)."""
    from relate_amd import io as rio
    rng = np.random.default_rng(5)
    post = rng.random((7, 3, 1234)) * 10.0 ** rng.uniform(-9, 0, (7, 3, 1234))
    post[:, :, rng.random(1234) < 0.05] = -1.0
    a, b = tmp_path / "a.post", tmp_path / "b.post"
    with open(a, "w") as fh:
        for n in range(len(post)):
            rio.write_post_rows(fh, post[n], k1_only)
    with open(b, "wb") as fh:
        rio.write_post_block(fh, post, k1_only)
    assert a.read_bytes() == b.read_bytes()
