"""Test infrastructure: ctypes binding of oracle/librelate_oracle.so (the plain-C CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_SO = os.path.join(ROOT, "oracle", "librelate_oracle.so")

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
u8p = C.POINTER(C.c_uint8)


def _d(a):
    return None if a is None else a.ctypes.data_as(dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(ip)


def _u8(a):
    return a.ctypes.data_as(u8p)


def build_if_missing():
    if not os.path.exists(ORC_SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])


class Oracle:
    def __init__(self):
        build_if_missing()
        L = self.L = C.CDLL(ORC_SO)
        L.orc_maf.argtypes = [u8p, C.c_int, C.c_int, dp]
        L.orc_maf_keep.restype = C.c_int
        L.orc_maf_keep.argtypes = [dp, C.c_int, C.c_double, ip]
        L.orc_ld_cell.argtypes = [u8p, u8p, C.c_int, C.c_int, dp]
        L.orc_hap_global.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp]
        L.orc_prune.restype = C.c_int
        L.orc_prune.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, ip]
        L.orc_err.restype = C.c_double
        L.orc_err.argtypes = [C.c_int, C.c_int, C.c_double]
        L.orc_emis_single.argtypes = [C.c_double, C.c_int, C.c_int, C.c_double, dp]
        L.orc_emis_ld.argtypes = [C.c_double] * 4 + [C.c_int] * 4 + [C.c_double, dp]
        L.orc_pair_stage.restype = C.c_int
        L.orc_pair_stage.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, C.c_double, C.c_int, C.c_int,
                                     C.c_int, dp, dp, dp, dp, dp, ip, dp, dp, dp, ip]
        L.orc_like.restype = C.c_double
        L.orc_like.argtypes = [dp, dp, dp, dp, dp, C.c_int, C.c_int, C.c_double]
        L.orc_decode.argtypes = [dp, dp, dp, dp, dp, C.c_int, C.c_int, C.c_int, dp]
        L.orc_viterbi.argtypes = [dp, dp, dp, dp, dp, C.c_int, C.c_int, ip]
        L.orc_calculateA.restype = C.c_double
        L.orc_calculateA.argtypes = [C.c_double] * 4
        L.orc_objective.restype = C.c_double
        L.orc_objective.argtypes = [C.c_int, dp, C.c_double, C.c_double, C.c_double, dp, dp, dp, dp, C.c_int, C.c_int,
                                    C.c_double]

    def maf(self, g):
        g = np.ascontiguousarray(g, dtype=np.uint8)
        out = np.zeros(g.shape[1])
        self.L.orc_maf(_u8(g), g.shape[0], g.shape[1], _d(out))
        return out

    def maf_keep(self, maf, mn):
        keep = np.zeros(len(maf), dtype=np.int32)
        self.L.orc_maf_keep(_d(np.ascontiguousarray(maf)), len(maf), mn, _i(keep))
        return keep

    def ld_cell(self, first, second):
        a = np.ascontiguousarray(first, dtype=np.uint8)
        b = np.ascontiguousarray(second, dtype=np.uint8)
        out = np.zeros(10)
        self.L.orc_ld_cell(_u8(a), _u8(b), len(a), 1, _d(out))
        return out

    def hap_global(self, g, depth, LD):
        g = np.ascontiguousarray(g, dtype=np.uint8)
        N, M = g.shape
        mea = np.zeros((depth, M - 1))
        tabs = [np.zeros((depth, M)) for _ in range(4)]
        self.L.orc_hap_global(_u8(g), N, M, depth, LD, _d(mea), *[_d(t) for t in tabs])
        return dict(mea=mea, pBA=tabs[0], pBa=tabs[1], pbA=tabs[2], pba=tabs[3])

    def prune(self, g, LD, back, prune_val):
        g = np.ascontiguousarray(g, dtype=np.uint8)
        keep = np.zeros(g.shape[1], dtype=np.int32)
        self.L.orc_prune(_u8(g), g.shape[0], g.shape[1], LD, back, prune_val, _i(keep))
        return keep

    def emis_single(self, p, o1, o2, eps):
        out = np.zeros(3)
        self.L.orc_emis_single(p, o1, o2, eps, _d(out))
        return out

    def emis_ld(self, pBA, pbA, pBa, pba, o1c, o2c, o1p, o2p, eps):
        out = np.zeros(3)
        self.L.orc_emis_ld(pBA, pbA, pBa, pba, o1c, o2c, o1p, o2p, eps, _d(out))
        return out

    def pair_stage(self, g, i, j, maf, pos, eps, use_ld, back, back2, hap):
        g = np.ascontiguousarray(g, dtype=np.uint8)
        N, M = g.shape
        keep = np.zeros(M, dtype=np.int32)
        t = np.zeros(M + 1)
        S1 = np.zeros((3, M))
        S = np.zeros((3, M))
        choose = np.zeros(M, dtype=np.int32)
        h = hap or {}
        m = self.L.orc_pair_stage(_u8(g), N, M, i, j, _d(np.ascontiguousarray(maf)), _d(np.ascontiguousarray(pos)),
                                  eps, use_ld, back, back2, _d(h.get("mea")), _d(h.get("pBA")), _d(h.get("pBa")),
                                  _d(h.get("pbA")), _d(h.get("pba")), _i(keep), _d(t), _d(S1), _d(S), _i(choose))
        return dict(m=m, keep=keep, t=t[:m + 1].copy(), S1=S1[:, :m].copy(), S=S[:, :m].copy(),
                    choose=choose[:m].copy())

    @staticmethod
    def logs(S):
        """(Sk1, Sk2, Sk3) = log of rows (2, 1, 0) as sgl forms them (src/relateHMM.cpp:3470-3472), with the C
        library's log as row_log has it (:410-423): numpy's vectorised log differs from it in the last bit for about
        one value in a thousand, which is enough to send a run of the chaotic 2-parameter fit somewhere else."""
        import math

        def lg(row):
            return np.array([math.log(v) if v > 0 else (-np.inf if v == 0 else np.nan) for v in row], dtype=np.float64)
        return lg(S[2]), lg(S[1]), lg(S[0])

    def like(self, delta, Sk1, Sk2, Sk3, t, double_recom, alim1):
        d = np.ascontiguousarray(delta, dtype=np.float64)
        return self.L.orc_like(_d(d), _d(Sk1), _d(Sk2), _d(Sk3), _d(t), len(Sk1), double_recom, alim1)

    def decode(self, par, Sk1, Sk2, Sk3, t, double_recom, use_k2=-1):
        p = np.ascontiguousarray(par, dtype=np.float64)
        if use_k2 < 0:
            use_k2 = int(p[1] != 0)
        post = np.zeros((3, len(Sk1)))
        self.L.orc_decode(_d(p), _d(Sk1), _d(Sk2), _d(Sk3), _d(t), len(Sk1), double_recom, use_k2, _d(post))
        return post

    def viterbi(self, par, Sk1, Sk2, Sk3, t, double_recom):
        p = np.ascontiguousarray(par, dtype=np.float64)
        path = np.zeros(len(Sk1), dtype=np.int32)
        self.L.orc_viterbi(_d(p), _d(Sk1), _d(Sk2), _d(Sk3), _d(t), len(Sk1), double_recom, _i(path))
        return path

    def calculateA(self, k0, k1, k2, phi):
        return self.L.orc_calculateA(k0, k1, k2, phi)

    def objective(self, mode, x, fixA_val, fixK2_val, phi, Sk1, Sk2, Sk3, t, double_recom, alim1):
        xx = np.ascontiguousarray(x, dtype=np.float64)
        return self.L.orc_objective(mode, _d(xx), fixA_val, fixK2_val, phi, _d(Sk1), _d(Sk2), _d(Sk3), _d(t),
                                    len(Sk1), double_recom, alim1)
