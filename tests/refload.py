"""Test infrastructure: ctypes bindings for the checkers under oracle/ and readers for the fixture data.

* ``Ref``    -- oracle/_ref/librelate_ref.so, the real reference compiled from its own sources
               (oracle/Makefile target ``ref``; harness oracle/ref_harness.cpp).
* data readers mirror the reference's text formats (src/filereader_and_conversions.cpp:31-94,317-480)
  for the fixture copies under tests/golden/data/.

Nothing here is imported by the product (relate_amd/).
"""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "data")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "librelate_ref.so")

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


def _d(a):
    return None if a is None else a.ctypes.data_as(dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(ip)


class RefOpts(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("min", "epsilon", "prune_val", "alim0", "alim1", "fixA_val", "fixK_val", "phi", "convTol")] + \
               [("par", C.c_double * 3)] + \
               [(n, C.c_int) for n in
                ("back", "back2", "doPrune", "LD", "ld_adj", "double_recom", "fixA", "fixK", "calcA",
                 "timesToConverge", "timesToRun", "doPar")]


def read_options(path):
    """27-line option file; first tab-separated token of each line parsed through *float*
    (src/filereader_and_conversions.cpp:32-48), then post-processed as src/runRelate.cpp:76-101."""
    vals = []
    with open(path) as fh:
        for line in fh:
            if len(vals) >= 27:
                break
            tok = line.split("\t")[0]
            mt = re.match(r"\s*[-+]?(\d+\.?\d*([eE][-+]?\d+)?|\.\d+([eE][-+]?\d+)?)", tok)  # atof semantics
            vals.append(float(np.float32(float(mt.group(0)))) if mt else 0.0)
    o = dict(
        doAllPairs=int(vals[0]), pair=(int(vals[1]), int(vals[2])), double_recom=int(vals[3]),
        LD=1 if int(vals[4]) == 0 else 0, min=vals[5], alim0=vals[6], alim1=vals[7], doPar=int(vals[8]),
        par=(vals[9], vals[10], vals[11]), ld_adj=int(vals[12]), epsilon=vals[13], back=int(vals[14]),
        doPrune=int(vals[15]), prune_val=vals[16], fixA=int(vals[17]), fixA_val=vals[18], fixK=int(vals[19]),
        fixK_val=vals[20], calcA=int(vals[21]), phi=vals[22], convTol=vals[23], timesToConverge=int(vals[24]),
        timesToRun=int(vals[25]), back2=int(vals[26]))
    if o["calcA"] and o["fixK"] and o["fixK_val"] == 0 and o["timesToRun"] != 1:
        o["timesToRun"] = 1
    o["back2"] = 2 * o["back"]
    return o


def read_geno(path):
    rows = []
    with open(path) as fh:
        for line in fh:
            toks = line.replace(",", " ").replace(";", " ").replace(":", " ").split()
            if toks:
                rows.append([0 if t in ("NA", "Na", "na") else int(t) for t in toks])
    return np.ascontiguousarray(np.array(rows, dtype=np.int32))


def read_vec(path, dtype):
    with open(path) as fh:
        toks = fh.read().replace(",", " ").replace(";", " ").replace(":", " ").split()
    return np.array([dtype(t) for t in toks])


def load_dataset(name):
    g = read_geno(os.path.join(DATA, name + ".geno"))
    pos = read_vec(os.path.join(DATA, name + ".pos"), float).astype(np.float64)
    chr_ = read_vec(os.path.join(DATA, name + ".chr"), int).astype(np.int32)
    order = np.lexsort((pos, chr_))  # mysort: (chr, pos) (src/asort.cpp:26-31)
    return np.ascontiguousarray(g[:, order]), np.ascontiguousarray(pos[order]), np.ascontiguousarray(chr_[order])


def load_plink(prefix):
    """doBimFile + bed_to_iMatrix + mysort (src/filereader_and_conversions.cpp:483-657, src/runRelate.cpp:363-381):
    autosomes 1..22 only, positions / 1e6, .bed codes recoded {00:3, 01:missing, 10:2, 11:1}."""
    chrs, poss = [], []
    with open(prefix + ".bim") as fh:
        for line in fh:
            t = line.split()
            if not t:
                break
            chrs.append(int(t[0]))
            poss.append(float(t[3]))
    with open(prefix + ".fam") as fh:
        n_ind = sum(1 for line in fh if line.strip())
    raw = np.fromfile(prefix + ".bed", dtype=np.uint8)
    assert raw[0] == 0x6C and raw[1] == 0x1B and raw[2] == 1, "SNP-major .bed expected"
    nsnp = len(chrs)
    bps = (n_ind + 3) // 4
    body = raw[3:3 + nsnp * bps].reshape(nsnp, bps)
    codes = np.stack([(body >> (2 * k)) & 3 for k in range(4)], axis=2).reshape(nsnp, bps * 4)[:, :n_ind]
    recode = np.array([3, 0, 2, 1], dtype=np.int32)  # recode {1,0,2,3} then 1 <-> 3 (:485, 530-533)
    g = recode[codes].T.copy()
    chrs = np.array(chrs, dtype=np.int32)
    poss = np.array(poss) / 1000000
    keep = (chrs != 0) & (chrs < 23)
    g, chrs, poss = g[:, keep], chrs[keep], poss[keep]
    order = np.lexsort((poss, chrs))
    return np.ascontiguousarray(g[:, order]), np.ascontiguousarray(poss[order]), np.ascontiguousarray(chrs[order])


def keep_pairs(path):
    """getTestIndividuals (src/filereader_and_conversions.cpp:447-474): all i<j among flagged individuals."""
    flags = read_vec(path, int)
    idx = [i for i, f in enumerate(flags) if f]
    return [(a, b) for ai, a in enumerate(idx) for b in idx[ai + 1:]]


def opts_struct(o):
    s = RefOpts()
    for k in ("min", "epsilon", "prune_val", "alim0", "alim1", "fixA_val", "fixK_val", "phi", "convTol", "back",
              "back2", "doPrune", "LD", "ld_adj", "double_recom", "fixA", "fixK", "calcA", "timesToConverge",
              "timesToRun", "doPar"):
        setattr(s, k, o[k])
    for i in range(3):
        s.par[i] = o["par"][i]
    return s


class Ref:
    """The compiled reference.  One session at a time (the reference keeps static globals)."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        L = self.L = C.CDLL(REF_SO)
        L.refh_like.restype = C.c_double
        L.refh_like.argtypes = [dp, dp, dp, dp, dp, C.c_int, C.c_int, dp]
        L.refh_decode.argtypes = [dp, dp, dp, dp, dp, C.c_int, C.c_int, C.c_int, dp]
        L.refh_viterbi.argtypes = [dp, dp, dp, dp, dp, C.c_int, C.c_int, ip]
        L.refh_emission_single.argtypes = [ip, ip, dp, C.c_int, C.c_double, dp]
        L.refh_ldemission.argtypes = [dp, dp, dp, dp, dp, C.c_int, dp]
        L.refh_error_ld.argtypes = [ip, ip, ip, ip, C.c_int, C.c_double, dp]
        L.refh_ld_cell.argtypes = [ip, ip, C.c_int, dp]
        L.refh_snp_pair_range.argtypes = [ip, C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp]
        L.refh_prune.restype = C.c_int
        L.refh_prune.argtypes = [ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, ip]
        L.refh_remove_cross.argtypes = [dp, C.c_int, C.c_int, ip, C.c_int, dp]
        L.refh_calculateA.restype = C.c_double
        L.refh_calculateA.argtypes = [C.c_double] * 4
        L.refh_calc_maf.argtypes = [ip, C.c_int, C.c_int, dp]
        L.refh_findmax_bfgs.restype = C.c_double
        L.refh_rand_stream.argtypes = [C.c_uint, C.c_int, ip]
        L.refh_init.restype = C.c_int
        L.refh_init.argtypes = [ip, C.c_int, C.c_int, dp, ip, C.POINTER(RefOpts)]
        L.refh_globals.argtypes = [ip, ip, dp, dp]
        L.refh_hap_global.restype = C.c_int
        L.refh_hap_global.argtypes = [dp] * 5
        L.refh_sgl.restype = C.c_int
        L.refh_sgl.argtypes = [C.c_int, C.c_int, C.c_long, dp, ip, dp, dp, dp, ip, dp, ip, dp]
        self._sess = None

    # -- kernels ---------------------------------------------------------------------------
    def like(self, delta, Sk1, Sk2, Sk3, t, double_recom, alim):
        d = np.ascontiguousarray(delta, dtype=np.float64)
        al = np.ascontiguousarray(alim, dtype=np.float64)
        return self.L.refh_like(_d(d), _d(Sk1), _d(Sk2), _d(Sk3), _d(t), len(Sk1), double_recom, _d(al))

    def decode(self, par, Sk1, Sk2, Sk3, t, double_recom, use_k2=-1):
        p = np.ascontiguousarray(par, dtype=np.float64)
        post = np.zeros((3, len(Sk1)))
        self.L.refh_decode(_d(p), _d(Sk1), _d(Sk2), _d(Sk3), _d(t), len(Sk1), double_recom, use_k2, _d(post))
        return post

    def viterbi(self, par, Sk1, Sk2, Sk3, t, double_recom):
        p = np.ascontiguousarray(par, dtype=np.float64)
        path = np.zeros(len(Sk1), dtype=np.int32)
        self.L.refh_viterbi(_d(p), _d(Sk1), _d(Sk2), _d(Sk3), _d(t), len(Sk1), double_recom, _i(path))
        return path

    def emission_single(self, ind1, ind2, maf, eps):
        i1 = np.ascontiguousarray(ind1, dtype=np.int32)
        i2 = np.ascontiguousarray(ind2, dtype=np.int32)
        mf = np.ascontiguousarray(maf, dtype=np.float64)
        out = np.zeros((3, len(i1)))
        self.L.refh_emission_single(_i(i1), _i(i2), _d(mf), len(i1), eps, _d(out))
        return out

    def ldemission(self, pBA, pbA, pBa, pba, ep):
        a, b, c, d = (np.ascontiguousarray(x, dtype=np.float64) for x in (pBA, pbA, pBa, pba))
        e = np.ascontiguousarray(ep, dtype=np.float64)
        out = np.zeros((3, len(a)))
        self.L.refh_ldemission(_d(a), _d(b), _d(c), _d(d), _d(e), len(a), _d(out))
        return out

    def error_ld(self, cur1, cur2, prev1, prev2, eps):
        arrs = [np.ascontiguousarray(x, dtype=np.int32) for x in (cur1, cur2, prev1, prev2)]
        out = np.zeros((81, len(arrs[0])))
        self.L.refh_error_ld(*[_i(a) for a in arrs], len(arrs[0]), eps, _d(out))
        return out

    def ld_cell(self, first, second):
        a = np.ascontiguousarray(first, dtype=np.int32)
        b = np.ascontiguousarray(second, dtype=np.int32)
        out = np.zeros(10)
        self.L.refh_ld_cell(_i(a), _i(b), len(a), _d(out))
        return out

    def snp_pair_range(self, geno, depth):
        g = np.ascontiguousarray(geno, dtype=np.int32)
        N, M = g.shape
        outs = [np.zeros((depth, M - 1)) for _ in range(6)]
        self.L.refh_snp_pair_range(_i(g), N, M, depth, *[_d(o) for o in outs])
        return dict(zip(("pBA", "pBa", "pbA", "pba", "D", "rsq2"), outs))

    def prune(self, geno, LD, back, prune_val):
        g = np.ascontiguousarray(geno, dtype=np.int32)
        keep = np.zeros(g.shape[1], dtype=np.int32)
        self.L.refh_prune(_i(g), g.shape[0], g.shape[1], LD, back, prune_val, _i(keep))
        return keep

    def remove_cross(self, mea, keep, back):
        m = np.ascontiguousarray(mea, dtype=np.float64)
        k = np.ascontiguousarray(keep, dtype=np.int32)
        out = np.zeros((back, int((k != 0).sum()) - 1))
        self.L.refh_remove_cross(_d(m), m.shape[0], m.shape[1], _i(k), back, _d(out))
        return out

    def calculateA(self, k0, k1, k2, phi):
        return self.L.refh_calculateA(k0, k1, k2, phi)

    def calc_maf(self, geno):
        g = np.ascontiguousarray(geno, dtype=np.int32)
        out = np.zeros(g.shape[1])
        self.L.refh_calc_maf(_i(g), g.shape[0], g.shape[1], _d(out))
        return out

    def findmax_bfgs(self, fun, x0, lo, hi):
        FT = C.CFUNCTYPE(C.c_double, dp)
        n = len(x0)
        cb = FT(lambda p: float(fun(np.array([p[i] for i in range(n)]))))
        x = np.array(x0, dtype=np.float64)
        l = np.array(lo, dtype=np.float64)
        h = np.array(hi, dtype=np.float64)
        self.L.refh_findmax_bfgs.argtypes = [C.c_int, dp, FT, dp, dp]
        r = self.L.refh_findmax_bfgs(n, _d(x), cb, _d(l), _d(h))
        return x, r

    def rand_stream(self, seed, n):
        out = np.zeros(n, dtype=np.int32)
        self.L.refh_rand_stream(seed, n, _i(out))
        return out

    # -- session ---------------------------------------------------------------------------
    def init(self, geno, pos, chr_, o):
        g = np.ascontiguousarray(geno, dtype=np.int32)
        p = np.ascontiguousarray(pos, dtype=np.float64)
        c = np.ascontiguousarray(chr_, dtype=np.int32)
        s = opts_struct(o)
        M = self.L.refh_init(_i(g), g.shape[0], g.shape[1], _d(p), _i(c), C.byref(s))
        if M < 0:
            raise RuntimeError("reference session already open")
        self._sess = dict(N=g.shape[0], Mraw=g.shape[1], M=M, o=o)
        return M

    def globals(self):
        s = self._sess
        keep = np.zeros(s["Mraw"], dtype=np.int32)
        genos = np.zeros((s["N"], s["M"]), dtype=np.int32)
        maf = np.zeros(s["M"])
        pos = np.zeros(s["M"])
        self.L.refh_globals(_i(keep), _i(genos), _d(maf), _d(pos))
        return keep, genos, maf, pos

    def hap_global(self):
        s = self._sess
        B2, M = s["o"]["back2"], s["M"]
        mea = np.zeros((B2, M - 1))
        tabs = [np.zeros((B2, M)) for _ in range(4)]
        self.L.refh_hap_global(_d(mea), *[_d(t) for t in tabs])
        return dict(mea=mea, pBA=tabs[0], pBa=tabs[1], pbA=tabs[2], pba=tabs[3])

    def sgl(self, i, j, seed=1234567):
        s = self._sess
        M, R = s["M"], max(1, s["o"]["timesToRun"])
        scal = np.zeros(10)
        keep = np.zeros(M, dtype=np.int32)
        t = np.zeros(M + 1)
        S1 = np.zeros(3 * M)
        Sm = np.zeros(3 * M)
        choose = np.zeros(M, dtype=np.int32)
        post = np.zeros(3 * M)
        path = np.zeros(M, dtype=np.int32)
        conv = np.zeros((R, 4))
        m = self.L.refh_sgl(i, j, seed, _d(scal), _i(keep), _d(t), _d(S1), _d(Sm), _i(choose), _d(post), _i(path),
                            _d(conv))
        return dict(a=scal[0], k2=scal[1], k1=scal[2], k0=scal[3], nll=scal[4], uLike=scal[5], poLike=scal[6],
                    timesRun=int(scal[7]), timesConv=int(scal[8]), m=m, keep=keep, t=t[:m + 1].copy(),
                    S1=S1[:3 * m].reshape(3, m).copy(), S=Sm[:3 * m].reshape(3, m).copy(), choose=choose[:m].copy(),
                    post=post[:3 * m].reshape(3, m).copy(), path=path[:m].copy(), conv=conv)

    def fini(self):
        self.L.refh_fini()
        self._sess = None
