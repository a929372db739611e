"""ctypes binding of librelate_hip.so (include/relate_hip.h) -- plumbing for tests, bench.py and the
multi-GPU launcher.  The product is the shared library; this module adds no numerics and has no CPU
fallback: if the library is missing or no gfx950 device is visible, it raises.

The method names mirror the reference's per-pair operator (relateHMM::init_globals / sgl,
src/relateHMM.h:58-79): ``Relate.init_globals(...)`` then ``like`` / ``fit`` / ``decode`` / ``sgl``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librelate_hip.so")

SYS_MAX = 10000000.0

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)
i8p = C.POINTER(C.c_int8)


class RelateOpts(C.Structure):
    """relate_opts of include/relate_hip.h (scalars of functionPars, src/types.h:232-276)."""
    _fields_ = [("min", C.c_double), ("epsilon", C.c_double), ("prune_val", C.c_double), ("alim", C.c_double * 2),
                ("fixA_val", C.c_double), ("fixK_val", C.c_double), ("phi", C.c_double), ("convTol", C.c_double),
                ("par", C.c_double * 3), ("back", C.c_int), ("doPrune", C.c_int), ("LD", C.c_int),
                ("ld_adj", C.c_int), ("double_recom", C.c_int), ("fixA", C.c_int), ("fixK", C.c_int),
                ("calcA", C.c_int), ("timesToConverge", C.c_int), ("timesToRun", C.c_int), ("doPar", C.c_int),
                ("seed_mode", C.c_int), ("seed", C.c_uint)]


class LinkageOpts(C.Structure):
    _fields_ = [("nsim", C.c_int), ("ccAll", C.c_int), ("infer", C.c_int), ("isK0", C.c_int), ("sig", C.c_float),
                ("seed", C.c_uint)]


class LinkageResult(C.Structure):
    _fields_ = [("pmaxsim", C.c_double), ("psim_over_nsim", C.c_double), ("psigsim", C.c_double),
                ("n_case_pairs", C.c_int), ("n_control_pairs", C.c_int)]


class RelateFit(C.Structure):
    _fields_ = [("a", C.c_double), ("k2", C.c_double), ("k1", C.c_double), ("k0", C.c_double),
                ("negloglik", C.c_double), ("uLike", C.c_double), ("poLike", C.c_double), ("timesRun", C.c_int),
                ("timesConverged", C.c_int), ("nsnp", C.c_int), ("status", C.c_int)]


def default_opts():
    """setDefaultParsRun (src/runRelate.cpp:104-147)."""
    return dict(min=0.0, epsilon=0.01, prune_val=0.2, alim0=0.001, alim1=0.15, fixA_val=0.0, fixK_val=0.0, phi=0.013,
                convTol=0.1, par=(0.0, 0.0, 0.0), back=5, doPrune=0, LD=1, ld_adj=1, double_recom=0, fixA=0, fixK=1,
                calcA=1, timesToConverge=2, timesToRun=3, doPar=0, seed_mode=0, seed=1234567)


def make_opts(o):
    d = default_opts()
    d.update({k: v for k, v in o.items() if k in d})
    s = RelateOpts()
    for k in ("min", "epsilon", "prune_val", "fixA_val", "fixK_val", "phi", "convTol", "back", "doPrune", "LD",
              "ld_adj", "double_recom", "fixA", "fixK", "calcA", "timesToConverge", "timesToRun", "doPar", "seed_mode",
              "seed"):
        setattr(s, k, d[k])
    s.alim[0], s.alim[1] = d["alim0"], d["alim1"]
    for i in range(3):
        s.par[i] = d["par"][i]
    return s


_lib = None


def load_library():
    """Load librelate_hip.so; raises (no fallback) if it has not been built (see __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("librelate_hip.so is not built at %s: run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: torch ships its own copy of libamdhip64 / libhsa-runtime64.  Loaded after this
    # library's copy it becomes a second HSA client of the process, and its initialisation then fails ("No HIP GPUs
    # are available") once this library holds page-locked result buffers.  With torch's copy loaded first the
    # library's NEEDED entry resolves to that same runtime, which is also what sharing device pointers with torch
    # (bench.py, synth.py) requires.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.relate_hip_create.argtypes = [C.POINTER(vp), C.c_int]
    L.relate_hip_destroy.argtypes = [vp]
    L.relate_hip_last_error.restype = C.c_char_p
    L.relate_hip_last_error.argtypes = [vp]
    L.relate_hip_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.relate_hip_host_free.argtypes = [vp]
    L.relate_post_text_stride.restype = C.c_size_t
    L.relate_post_text_stride.argtypes = [C.c_int]
    L.relate_format_post_rows.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_size_t)]
    L.relate_hip_load.argtypes = [vp, u8p, C.c_int, C.c_int, dp, ip, C.POINTER(RelateOpts)]
    L.relate_hip_load_bed.argtypes = [vp, u8p, C.c_int, C.c_int, ip, C.c_int, dp, ip, C.POINTER(RelateOpts)]
    L.relate_hip_precalc.argtypes = [vp, ip, ip]
    L.relate_hip_get_stripped.argtypes = [vp, u8p, dp, dp]
    L.relate_hip_get_hap.argtypes = [vp, dp, dp, dp, dp, dp, ip]
    L.relate_hip_pair_emissions.argtypes = [vp, C.c_int, C.c_int, ip, dp, dp]
    L.relate_hip_like.argtypes = [vp, i32p, C.c_int, dp, C.c_int, dp]
    L.relate_hip_like_noalloc.argtypes = [vp, C.c_int, vp, C.c_int, vp]
    L.relate_hip_fit.argtypes = [vp, i32p, C.c_int, dp, C.POINTER(RelateFit), dp]
    L.relate_hip_decode.argtypes = [vp, i32p, C.c_int, dp, dp, i8p, ip]
    L.relate_hip_decode_rows.argtypes = [vp, i32p, C.c_int, dp, C.c_int, dp, ip]
    L.relate_format_rows.argtypes = [dp, C.c_long, C.c_int, C.c_char_p, C.POINTER(C.c_size_t)]
    L.relate_hip_bench_decode_rows.argtypes = [vp, i32p, C.c_int, dp, C.c_int, C.c_int, dp, dp]
    L.relate_hip_decode_f32.argtypes = [vp, i32p, C.c_int, dp, C.c_int, C.POINTER(C.c_float), ip]
    L.relate_hip_run_pairs.argtypes = [vp, i32p, C.c_int, dp, C.POINTER(RelateFit), dp, i8p]
    L.relate_rand_stream.argtypes = [C.c_uint, C.c_int, ip]
    L.relate_rand_stream.restype = None
    L.relate_hip_shard.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.relate_hip_shard.restype = None
    L.relate_hip_last_timings.argtypes = [vp, dp]
    L.relate_hip_last_chunks.argtypes = [vp, ip, ip]
    L.relate_hip_last_like_stats.argtypes = [vp, dp]
    L.relate_hip_last_gather_ranks.argtypes = [vp]
    L.relate_hip_gather.argtypes = [vp, C.POINTER(RelateFit), C.c_int64, C.c_int, C.c_int, C.c_char_p, C.c_int,
                                    C.POINTER(RelateFit)]
    fpt = C.POINTER(C.c_float)
    L.relate_hip_linkage.argtypes = [vp, fpt, C.c_int, C.c_int, fpt, fpt, ip, C.c_int, C.POINTER(LinkageOpts), fpt, fpt,
                                     fpt, fpt, C.POINTER(LinkageResult)]
    L.relate_hip_linkage_pairs.argtypes = [vp, i32p, C.c_int, dp, C.c_int, fpt, fpt, ip, C.c_int, C.POINTER(LinkageOpts),
                                           fpt, fpt, fpt, fpt, C.POINTER(LinkageResult)]
    L.relate_hip_bench_like.argtypes = [vp, i32p, C.c_int, dp, C.c_int, C.c_int, dp]
    L.relate_hip_bench_decode.argtypes = [vp, i32p, C.c_int, dp, C.c_int, dp, dp]
    _lib = L
    return L


FINDMAX_CB = C.CFUNCTYPE(C.c_double, dp, C.c_void_p)


def findmax_bfgs(fun, x0, lo, hi):
    """findmax_bfgs (src/bfgs.h:63-66) on a Python callback; returns (x, -f)."""
    L = load_library()
    n = len(x0)
    cb = FINDMAX_CB(lambda p, _u: float(fun(np.array([p[i] for i in range(n)]))))
    L.relate_findmax_bfgs.argtypes = [C.c_int, dp, FINDMAX_CB, C.c_void_p, dp, dp, dp]
    x = np.array(x0, dtype=np.float64)
    l = np.array(lo, dtype=np.float64)
    h = np.array(hi, dtype=np.float64)
    nf = C.c_double(0)
    rc = L.relate_findmax_bfgs(n, x.ctypes.data_as(dp), cb, None, l.ctypes.data_as(dp), h.ctypes.data_as(dp),
                               C.byref(nf))
    if rc:
        raise RuntimeError("relate_findmax_bfgs failed: %d" % rc)
    return x, nf.value


def rand_stream(seed, n):
    L = load_library()
    out = np.zeros(n, dtype=np.int32)
    L.relate_rand_stream(seed, n, out.ctypes.data_as(ip))
    return out


def gather_records(handle, mine, P_total, rank, nranks, rendezvous, transport=0):
    """relate_hip_gather: `mine` = this rank's RelateFit records (a ctypes array, or a structured numpy array of
    fit_dtype()), in the order of its relate_hip_shard range; returns all P_total records as a ctypes array.
    handle: a Relate (its GPU and stream carry the RCCL all-gather) or None (file transport only)."""
    L = load_library()
    if isinstance(mine, np.ndarray):
        buf = (RelateFit * max(len(mine), 1)).from_buffer_copy(
            np.ascontiguousarray(mine).tobytes().ljust(C.sizeof(RelateFit), b"\0"))
    else:
        buf = mine
    out = (RelateFit * max(int(P_total), 1))()
    h = handle.h if handle is not None else None
    rc = L.relate_hip_gather(h, buf, int(P_total), rank, nranks,
                             None if rendezvous is None else os.fsencode(rendezvous), transport, out)
    if rc:
        raise RelateError("relate_hip_gather: %s" % L.relate_hip_last_error(h).decode())
    return out


def fit_dtype():
    """numpy view of relate_fit (include/relate_hip.h)."""
    return np.dtype([(n, np.float64 if t is C.c_double else np.int32) for n, t in RelateFit._fields_])


def shard(P, rank, nranks):
    L = load_library()
    b, e = C.c_int64(0), C.c_int64(0)
    L.relate_hip_shard(P, rank, nranks, C.byref(b), C.byref(e))
    return b.value, e.value


class RelateError(RuntimeError):
    pass


def pinned_empty(shape, dtype=np.float64):
    """numpy array in page-locked host memory (relate_hip_host_alloc): posterior rows and paths written into it by
    Relate.decode(out=...) arrive by direct DMA instead of through the library's staging buffers.  Freed when the
    array (and every view of it) is gone."""
    import weakref
    L = load_library()
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    ptr = C.c_void_p()
    if L.relate_hip_host_alloc(C.byref(ptr), n):
        raise RelateError("relate_hip_host_alloc: %s" % L.relate_hip_last_error(None).decode())
    buf = (C.c_char * max(n, 1)).from_address(ptr.value)
    weakref.finalize(buf, L.relate_hip_host_free, C.c_void_p(ptr.value))
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


class Relate:
    """One handle = one GPU.  Mirrors relateHMM: init_globals() once, then batched sgl pieces."""

    def __init__(self, device=0):
        self.L = load_library()
        self.h = C.c_void_p()
        rc = self.L.relate_hip_create(C.byref(self.h), device)
        if rc:
            raise RelateError("relate_hip_create: %s" % self.L.relate_hip_last_error(None).decode())
        self.M = None
        self.N = None
        self.opts = None

    def _chk(self, rc):
        if rc:
            raise RelateError(self.L.relate_hip_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.L.relate_hip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- init_globals = load + pre_calc ------------------------------------------------------------
    def init_globals(self, geno, pos, chr_, opts):
        g = np.ascontiguousarray(geno, dtype=np.uint8)
        p = np.ascontiguousarray(pos, dtype=np.float64)
        c = None if chr_ is None else np.ascontiguousarray(chr_, dtype=np.int32)
        self.opts = dict(default_opts(), **{k: v for k, v in opts.items()})
        s = make_opts(self.opts)
        self.N, self.Mraw = g.shape
        self._chk(self.L.relate_hip_load(self.h, g.ctypes.data_as(u8p), g.shape[0], g.shape[1], p.ctypes.data_as(dp),
                                         None if c is None else c.ctypes.data_as(ip), C.byref(s)))
        keep = np.zeros(self.Mraw, dtype=np.int32)
        M = C.c_int(0)
        self._chk(self.L.relate_hip_precalc(self.h, keep.ctypes.data_as(ip), C.byref(M)))
        self.M = M.value
        self.keep = keep
        return self.M

    def init_globals_bed(self, bed_payload, n_ind, cols, pos, chr_, opts):
        """init_globals on a PLINK .bed payload (SNP-major, header stripped): the 2-bit fields are recoded on the GPU,
        no byte-per-genotype matrix is formed.  cols/pos/chr_: relate_amd.io.read_plink_raw()."""
        b = np.ascontiguousarray(bed_payload, dtype=np.uint8)
        cl = np.ascontiguousarray(cols, dtype=np.int32)
        p = np.ascontiguousarray(pos, dtype=np.float64)
        c = None if chr_ is None else np.ascontiguousarray(chr_, dtype=np.int32)
        bps = (n_ind + 3) // 4
        if b.size % bps:
            raise ValueError(".bed payload is not a whole number of %d-byte SNP rows" % bps)
        self.opts = dict(default_opts(), **{k: v for k, v in opts.items()})
        s = make_opts(self.opts)
        self.N, self.Mraw = n_ind, len(cl)
        self._chk(self.L.relate_hip_load_bed(self.h, b.ctypes.data_as(u8p), n_ind, b.size // bps, cl.ctypes.data_as(ip),
                                             len(cl), p.ctypes.data_as(dp),
                                             None if c is None else c.ctypes.data_as(ip), C.byref(s)))
        keep = np.zeros(self.Mraw, dtype=np.int32)
        M = C.c_int(0)
        self._chk(self.L.relate_hip_precalc(self.h, keep.ctypes.data_as(ip), C.byref(M)))
        self.M = M.value
        self.keep = keep
        return self.M

    def stripped(self, want_geno=True):
        g = np.zeros((self.N, self.M), dtype=np.uint8) if want_geno else None
        pos = np.zeros(self.M)
        maf = np.zeros(self.M)
        self._chk(self.L.relate_hip_get_stripped(self.h, None if g is None else g.ctypes.data_as(u8p),
                                                 pos.ctypes.data_as(dp), maf.ctypes.data_as(dp)))
        return g, pos, maf

    def hap_global(self):
        B2 = 2 * self.opts["back"]
        mea = np.zeros((B2, self.M - 1))
        t = [np.zeros((B2, self.M)) for _ in range(4)]
        ch = np.zeros(self.M, dtype=np.int32)
        self._chk(self.L.relate_hip_get_hap(self.h, mea.ctypes.data_as(dp), *[x.ctypes.data_as(dp) for x in t],
                                            ch.ctypes.data_as(ip)))
        return dict(mea=mea, pBA=t[0], pBa=t[1], pbA=t[2], pba=t[3], choose=ch)

    def pair_emissions(self, i, j):
        keep = np.zeros(self.M, dtype=np.int32)
        t = np.zeros(self.M)
        S = np.zeros((3, self.M))
        self._chk(self.L.relate_hip_pair_emissions(self.h, i, j, keep.ctypes.data_as(ip), t.ctypes.data_as(dp),
                                                   S.ctypes.data_as(dp)))
        return keep, t, S

    @staticmethod
    def _pairs(pairs):
        p = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        return p, p.ctypes.data_as(i32p)

    # -- batched kernels ---------------------------------------------------------------------------
    def like(self, pairs, params):
        """params: P x E x 3 (a,k2,k1) -> P x E negative log-likelihoods (like(), src/relateHMM.cpp:1426)."""
        p, pp = self._pairs(pairs)
        par = np.ascontiguousarray(params, dtype=np.float64)
        if len(p) == 0:
            return np.zeros((0, par.shape[1] if par.ndim == 3 else 0))
        par = par.reshape(len(p), -1, 3)
        out = np.zeros((len(p), par.shape[1]))
        self._chk(self.L.relate_hip_like(self.h, pp, len(p), par.ctypes.data_as(dp), par.shape[1],
                                         out.ctypes.data_as(dp)))
        return out

    def like_noalloc(self, P, d_params_ptr, E, d_out_ptr):
        """relate_hip_like_noalloc on the block staged by the previous call; device addresses in and out."""
        self._chk(self.L.relate_hip_like_noalloc(self.h, P, C.c_void_p(d_params_ptr), E, C.c_void_p(d_out_ptr)))

    def fit(self, pairs, starts=None, want_conv=False):
        p, pp = self._pairs(pairs)
        R = max(1, self.opts["timesToRun"])
        out = (RelateFit * len(p))()
        st = None if starts is None else np.ascontiguousarray(starts, dtype=np.float64).reshape(len(p), R, 4)
        conv = np.zeros((len(p), R, 4)) if want_conv else None
        self._chk(self.L.relate_hip_fit(self.h, pp, len(p), None if st is None else st.ctypes.data_as(dp), out,
                                        None if conv is None else conv.ctypes.data_as(dp)))
        res = [{f: getattr(o, f) for f, _ in RelateFit._fields_} for o in out]
        return (res, conv) if want_conv else res

    def decode(self, pairs, params, want_post=True, want_path=False, out=None):
        """out: optional float64 buffer of at least len(pairs) x 3 x M values (e.g. from pinned_empty) that receives
        the posteriors; the returned array is a view of it."""
        p, pp = self._pairs(pairs)
        par = np.ascontiguousarray(params, dtype=np.float64).reshape(len(p), 3)
        if out is not None and want_post:
            flat = out.reshape(-1)
            if flat.dtype != np.float64 or not flat.flags.c_contiguous or flat.size < len(p) * 3 * self.M:
                raise ValueError("out must be a contiguous float64 buffer of at least len(pairs) * 3 * M values")
            post = flat[:len(p) * 3 * self.M].reshape(len(p), 3, self.M)
        else:
            post = np.empty((len(p), 3, self.M)) if want_post else None
        path = np.zeros((len(p), self.M), dtype=np.int8) if want_path else None
        ns = np.zeros(len(p), dtype=np.int32)
        self._chk(self.L.relate_hip_decode(self.h, pp, len(p), par.ctypes.data_as(dp),
                                           None if post is None else post.ctypes.data_as(dp),
                                           None if path is None else path.ctypes.data_as(i8p), ns.ctypes.data_as(ip)))
        return post, path, ns

    def decode_rows(self, pairs, params, rows=2, out=None):
        """Posterior rows in the .post writer's order (IBD0, IBD1[, IBD2]) as float64: (len(pairs), rows, M); the
        values of decode(), only the rows asked for (relate_hip_decode_rows)."""
        p, pp = self._pairs(pairs)
        par = np.ascontiguousarray(params, dtype=np.float64).reshape(len(p), 3)
        n = len(p) * rows * self.M
        if out is not None:
            flat = out.reshape(-1)
            if flat.dtype != np.float64 or not flat.flags.c_contiguous or flat.size < n:
                raise ValueError("out must be a contiguous float64 buffer of at least len(pairs) * rows * M values")
            post = flat[:n].reshape(len(p), rows, self.M)
        else:
            post = np.empty((len(p), rows, self.M))
        ns = np.zeros(len(p), dtype=np.int32)
        self._chk(self.L.relate_hip_decode_rows(self.h, pp, len(p), par.ctypes.data_as(dp), rows,
                                                post.ctypes.data_as(dp), ns.ctypes.data_as(ip)))
        return post, ns

    def decode_f32(self, pairs, params, rows=2, out=None):
        """Posterior rows in the .post writer's order (IBD0, IBD1[, IBD2]) as float32: (len(pairs), rows, M).
        out: optional float32 buffer (e.g. pinned_empty(..., np.float32)) that receives them."""
        p, pp = self._pairs(pairs)
        par = np.ascontiguousarray(params, dtype=np.float64).reshape(len(p), 3)
        n = len(p) * rows * self.M
        if out is not None:
            flat = out.reshape(-1)
            if flat.dtype != np.float32 or not flat.flags.c_contiguous or flat.size < n:
                raise ValueError("out must be a contiguous float32 buffer of at least len(pairs) * rows * M values")
            post = flat[:n].reshape(len(p), rows, self.M)
        else:
            post = np.empty((len(p), rows, self.M), dtype=np.float32)
        ns = np.zeros(len(p), dtype=np.int32)
        self._chk(self.L.relate_hip_decode_f32(self.h, pp, len(p), par.ctypes.data_as(dp), rows,
                                               post.ctypes.data_as(C.POINTER(C.c_float)), ns.ctypes.data_as(ip)))
        return post, ns

    def post_buffer(self, npairs, rows=3, dtype=np.float64):
        """Page-locked buffer for the posteriors of npairs pairs (decode's `out`)."""
        return pinned_empty((npairs, rows, self.M), dtype)

    def sgl(self, pairs, starts=None, want_post=True, want_path=False):
        """Body of the reference's pair loop (src/relateHMM.cpp:3764-3785): fit then decode at the estimate."""
        p, pp = self._pairs(pairs)
        R = max(1, self.opts["timesToRun"])
        out = (RelateFit * len(p))()
        st = None if starts is None else np.ascontiguousarray(starts, dtype=np.float64).reshape(len(p), R, 4)
        post = np.zeros((len(p), 3, self.M)) if want_post else None
        path = np.zeros((len(p), self.M), dtype=np.int8) if want_path else None
        self._chk(self.L.relate_hip_run_pairs(self.h, pp, len(p), None if st is None else st.ctypes.data_as(dp), out,
                                              None if post is None else post.ctypes.data_as(dp),
                                              None if path is None else path.ctypes.data_as(i8p)))
        res = [{f: getattr(o, f) for f, _ in RelateFit._fields_} for o in out]
        return res, post, path

    # -- linkage (HMMtest) -------------------------------------------------------------------------
    def linkage(self, post, cc, pos=None, K=None, isK0=0, nsim=10, ccAll=0, sig=0.05, infer=1, seed=1,
                want_cov=False, want_post=False):
        """uber_main of the reference's HMMtest (src/HMMtest.txt:1195-1367), case-control design."""
        fpt = C.POINTER(C.c_float)
        p = np.ascontiguousarray(post, dtype=np.float32)
        R, M = p.shape
        c = np.ascontiguousarray(cc, dtype=np.int32)
        ps = None if pos is None else np.ascontiguousarray(pos, dtype=np.float32)
        k = None if K is None else np.ascontiguousarray(K, dtype=np.float32)
        o = LinkageOpts(nsim, ccAll, infer, isK0, sig, seed)
        stat = np.zeros(M, dtype=np.float32)
        xmax = np.zeros(nsim, dtype=np.float32)
        cov = np.zeros((R, R), dtype=np.float32) if want_cov else None
        pout = np.zeros((R, M), dtype=np.float32) if want_post else None
        res = LinkageResult()
        f = lambda a: None if a is None else a.ctypes.data_as(fpt)  # noqa: E731
        self._chk(self.L.relate_hip_linkage(self.h, f(p), R, M, f(ps), f(k), c.ctypes.data_as(ip), len(c), C.byref(o),
                                            f(stat), f(xmax), f(cov), f(pout), C.byref(res)))
        return dict(stat=stat, xmax=xmax, cov=cov, post=pout, pmaxsim=res.pmaxsim, psim_over_nsim=res.psim_over_nsim,
                    psigsim=res.psigsim, n_case_pairs=res.n_case_pairs, n_control_pairs=res.n_control_pairs)

    def linkage_pairs(self, pairs, params, cc, digits=4, pos=None, K=None, isK0=0, nsim=10, ccAll=0, sig=0.05, infer=1,
                      seed=1, want_cov=False, want_post=False):
        """Decode (IBD0 row) + HMMtest without the posterior rows leaving the device (relate_hip_linkage_pairs)."""
        fpt = C.POINTER(C.c_float)
        p, pp = self._pairs(pairs)
        par = np.ascontiguousarray(params, dtype=np.float64).reshape(len(p), 3)
        R, M = len(p), self.M
        c = np.ascontiguousarray(cc, dtype=np.int32)
        ps = None if pos is None else np.ascontiguousarray(pos, dtype=np.float32)
        k = None if K is None else np.ascontiguousarray(K, dtype=np.float32)
        o = LinkageOpts(nsim, ccAll, infer, isK0, sig, seed)
        stat = np.zeros(M, dtype=np.float32)
        xmax = np.zeros(nsim, dtype=np.float32)
        cov = np.zeros((R, R), dtype=np.float32) if want_cov else None
        pout = np.zeros((R, M), dtype=np.float32) if want_post else None
        res = LinkageResult()
        f = lambda a: None if a is None else a.ctypes.data_as(fpt)  # noqa: E731
        self._chk(self.L.relate_hip_linkage_pairs(self.h, pp, R, par.ctypes.data_as(dp), digits, f(ps), f(k),
                                                  c.ctypes.data_as(ip), len(c), C.byref(o), f(stat), f(xmax), f(cov),
                                                  f(pout), C.byref(res)))
        return dict(stat=stat, xmax=xmax, cov=cov, post=pout, pmaxsim=res.pmaxsim, psim_over_nsim=res.psim_over_nsim,
                    psigsim=res.psigsim, n_case_pairs=res.n_case_pairs, n_control_pairs=res.n_control_pairs)

    # -- measurement -------------------------------------------------------------------------------
    def timings(self):
        t = np.zeros(4)
        self.L.relate_hip_last_timings(self.h, t.ctypes.data_as(dp))
        return dict(stage_ms=t[0], like_ms=t[1], decode_fwd_ms=t[2], decode_bwd_ms=t[3])

    def like_stats(self):
        """Likelihood launches of the last fit: point-steps, kernel ms and launch counts by route (sweep, chunked)."""
        s = np.zeros(6)
        self.L.relate_hip_last_like_stats(self.h, s.ctypes.data_as(dp))
        return dict(sweep_point_steps=s[0], chunked_point_steps=s[1], sweep_ms=s[2], chunked_ms=s[3],
                    sweep_launches=int(s[4]), chunked_launches=int(s[5]))

    def gather_ranks(self):
        """Ranks the last relate_hip_gather on this handle spanned (ncclCommCount, or the file session's size)."""
        return int(self.L.relate_hip_last_gather_ranks(self.h))

    def last_chunks(self):
        a, b = C.c_int(0), C.c_int(0)
        self.L.relate_hip_last_chunks(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def bench_like(self, pairs, params, reps):
        p, pp = self._pairs(pairs)
        par = np.ascontiguousarray(params, dtype=np.float64).reshape(len(p), -1, 3)
        ms = C.c_double(0)
        self._chk(self.L.relate_hip_bench_like(self.h, pp, len(p), par.ctypes.data_as(dp), par.shape[1], reps,
                                               C.byref(ms)))
        return ms.value

    def bench_decode(self, pairs, params, reps, rows=0):
        """rows: 0 = the reference's three rows; 1..3 = the first rows of the writer's order (IBD0, IBD1, IBD2)."""
        p, pp = self._pairs(pairs)
        par = np.ascontiguousarray(params, dtype=np.float64).reshape(len(p), 3)
        a, b = C.c_double(0), C.c_double(0)
        self._chk(self.L.relate_hip_bench_decode_rows(self.h, pp, len(p), par.ctypes.data_as(dp), rows, reps,
                                                      C.byref(a), C.byref(b)))
        return a.value, b.value
