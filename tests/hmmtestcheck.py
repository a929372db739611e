"""Shared helpers of the linkage (HMMtest) tests: fixture readers in the reference's float semantics and the
ctypes binding of the CPU oracle (oracle/hmmtest_oracle.c)."""
import ctypes as C
import json
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)
dp = C.POINTER(C.c_double)
_SPLIT = re.compile(r"[;: \t\r\n]+")  # HMMtest's delimiters for the post / position files (no comma)


def golden():
    with open(os.path.join(GOLD, "hmmtest", "golden.json")) as fh:
        return json.load(fh)


def read_post(path):
    rows = []
    with open(path) as fh:
        for line in fh:
            t = [x for x in _SPLIT.split(line) if x]
            if t:
                rows.append(np.array(t, dtype=np.float64).astype(np.float32))  # stringstream >> float
    return np.ascontiguousarray(np.stack(rows))


def read_floats(path):
    with open(path) as fh:
        return np.array([x for x in _SPLIT.split(fh.read()) if x], dtype=np.float64).astype(np.float32)


def read_cc(path):
    with open(path) as fh:
        return np.array([int(x) for x in re.split(r"[;,: \t\r\n]+", fh.read()) if x], dtype=np.int32)


def case_inputs(case):
    """(post, pos or None, K or None, isK0, cc, nsim, ccAll, sig, infer) of a golden case from its argument list."""
    a = case["args"]
    g = lambda f, d=None: (a[a.index(f) + 1] if f in a else d)  # noqa: E731
    post = read_post(os.path.join(GOLD, g("-P")))
    pos = read_floats(os.path.join(GOLD, g("-p"))) if g("-p") else None
    K = read_floats(os.path.join(GOLD, g("-K"))) if g("-K") else None
    cc = read_cc(os.path.join(GOLD, g("-D")))
    return dict(post=post, pos=pos, K=K, isK0=0, cc=cc, nsim=int(g("-Nsim", 10)), ccAll=int(g("-ccAll", 0)), sig=0.05,
                infer=int(g("-infer", 1)))


def oracle_linkage(inp, seed):
    L = C.CDLL(os.path.join(ROOT, "oracle", "librelate_oracle.so"))
    L.orc_linkage.restype = C.c_int
    L.orc_linkage.argtypes = [fp, C.c_int, C.c_int, fp, fp, C.c_int, ip, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                              C.c_uint, fp, fp, fp, fp, dp]
    post = inp["post"].copy()
    R, M = post.shape
    K = np.zeros(R, dtype=np.float32)
    cov = np.zeros((R, R), dtype=np.float32)
    stat = np.zeros(M, dtype=np.float32)
    xmax = np.zeros(inp["nsim"], dtype=np.float32)
    summ = np.zeros(5)
    pos, Kin, cc = inp["pos"], inp["K"], inp["cc"]
    rc = L.orc_linkage(post.ctypes.data_as(fp), R, M, None if pos is None else pos.ctypes.data_as(fp),
                       None if Kin is None else Kin.ctypes.data_as(fp), inp["isK0"], cc.ctypes.data_as(ip), len(cc),
                       inp["nsim"], inp["ccAll"], inp["sig"], inp["infer"], seed, K.ctypes.data_as(fp),
                       cov.ctypes.data_as(fp), stat.ctypes.data_as(fp), xmax.ctypes.data_as(fp), summ.ctypes.data_as(dp))
    assert rc == 0
    return dict(post=post, K=K, cov=cov, stat=stat, xmax=xmax, pmaxsim=summ[0], psim_over_nsim=summ[1], psigsim=summ[2],
                n_case_pairs=int(summ[3]), n_control_pairs=int(summ[4]))


def as_printed(v):
    """ostream << float at the default precision 6."""
    return np.array([float("%.6g" % x) for x in np.asarray(v, dtype=np.float32)])
