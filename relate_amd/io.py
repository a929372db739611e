"""Input/output formats of the reference, for the Python-side drivers (relate_amd/allpairs.py): the 27-line option
file, whitespace/punctuation separated text matrices and vectors, the keep list, PLINK .bed/.bim/.fam, and the
.post / .k writers.  Same semantics as the C++ host driver relate_amd/csrc/host/relate_cli.cpp.
References: src/filereader_and_conversions.cpp:31-94 (options), :230-257, 317-480 (text), :447-474 (keep list),
:483-657 (PLINK), src/asort.cpp:37-88 (sort), src/relateHMM.cpp:117-144 (writers), src/runRelate.cpp:76-101."""
import re

import numpy as np

_SPLIT = re.compile(r"[,;: \t\r\n]+")
_NUM = re.compile(r"\s*[-+]?(\d+\.?\d*([eE][-+]?\d+)?|\.\d+([eE][-+]?\d+)?)")


def _f32(x):
    return float(np.float32(x))


def read_options(path):
    """First tab-separated token of each of the 27 lines, parsed like atof into a *float* (:32-48)."""
    vals = []
    with open(path) as fh:
        for line in fh:
            if len(vals) == 27:
                break
            m = _NUM.match(line.split("\t")[0])
            vals.append(_f32(float(m.group(0))) if m else 0.0)
    if len(vals) < 27:
        raise ValueError("options file %s has fewer than 27 lines" % path)
    o = dict(doAllPairs=int(vals[0]), pair=(int(vals[1]), int(vals[2])), double_recom=int(vals[3]),
             LD=1 if int(vals[4]) == 0 else 0, min=vals[5], alim0=vals[6], alim1=vals[7], doPar=int(vals[8]),
             par=(vals[9], vals[10], vals[11]), ld_adj=int(vals[12]), epsilon=vals[13], back=int(vals[14]),
             doPrune=int(vals[15]), prune_val=vals[16], fixA=int(vals[17]), fixA_val=vals[18], fixK=int(vals[19]),
             fixK_val=vals[20], calcA=int(vals[21]), phi=vals[22], convTol=vals[23], timesToConverge=int(vals[24]),
             timesToRun=int(vals[25]))
    if o["calcA"] and o["fixK"] and o["fixK_val"] == 0 and o["timesToRun"] != 1:
        o["timesToRun"] = 1  # a one-dimensional optimisation is run once (src/runRelate.cpp:77-81)
    if o["doPrune"] and o["back"] == 0:
        raise ValueError("Must supply a back > 0, when you want to prune")
    o["back2"] = 2 * o["back"]
    return o


def _tokens(path):
    with open(path) as fh:
        return [t for t in _SPLIT.split(fh.read()) if t]


def read_vector(path, dtype=float):
    return np.array([0 if t in ("NA", "Na", "na") else dtype(t) for t in _tokens(path)])


def read_geno(path):
    rows = []
    with open(path) as fh:
        for line in fh:
            t = [x for x in _SPLIT.split(line) if x]
            if t:
                rows.append([0 if x in ("NA", "Na", "na") else int(x) for x in t])
    n = len(rows[0])
    for r, row in enumerate(rows):
        if len(row) != n:
            raise ValueError("row length mismatch at line:%d" % (r + 1))
    g = np.array(rows, dtype=np.int64)
    if g.min() < 0 or g.max() > 3:
        raise ValueError("Error in genotype data: value should be between 0 and 3")
    return g.astype(np.uint8)


def sort_snps(geno, pos, chr_):
    """mysort: (chr, pos); equal neighbouring positions abort."""
    order = np.lexsort((pos, chr_))
    p = pos[order]
    if len(p) > 1 and (p[1:] == p[:-1]).any():
        raise ValueError("Non unique position")
    return np.ascontiguousarray(geno[:, order]), np.ascontiguousarray(p), np.ascontiguousarray(chr_[order])


def load_text(geno, pos, chr_=None):
    g = read_geno(geno)
    p = read_vector(pos, float).astype(np.float64)
    if chr_ is None:
        c = np.full(len(p), -1, dtype=np.int32)
        if g.shape[1] != len(p):
            raise ValueError("Dimension of data input doesn't have compatible dimensions")
        return g, p, c
    c = read_vector(chr_, int).astype(np.int32)
    if g.shape[1] != len(p) or len(c) != len(p):
        raise ValueError("Dimension of data input doesn't have compatible dimensions")
    return sort_snps(g, p, c)


def read_bim(path):
    """Chromosome and position columns of a .bim file (doBimFile, src/filereader_and_conversions.cpp:593-657): six
    columns per line, reading stops at the first empty line."""
    chrs, poss = [], []
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t:
                break
            if len(t) != 6:
                raise ValueError("plink bim file doesn't have 6 columns")
            chrs.append(int(t[0]))
            poss.append(float(t[3]))
    return chrs, poss


def load_plink(prefix):
    chrs, poss = read_bim(prefix + ".bim")
    with open(prefix + ".fam") as fh:
        n_ind = sum(1 for line in fh if line.strip())
    raw = np.fromfile(prefix + ".bed", dtype=np.uint8)
    if raw[0] != 0x6C or raw[1] != 0x1B:
        raise ValueError("Input file does not appear to be a .bed file")
    nsnp = len(chrs)
    recode = np.array([3, 0, 2, 1], dtype=np.uint8)  # {1,0,2,3} then 1 <-> 3 (:485, 530-533)
    if raw[2]:
        bps = (n_ind + 3) // 4
        body = raw[3:3 + nsnp * bps].reshape(nsnp, bps)
        codes = np.stack([(body >> (2 * k)) & 3 for k in range(4)], axis=2).reshape(nsnp, bps * 4)[:, :n_ind].T
    else:
        bps = (nsnp + 3) // 4
        body = raw[3:3 + n_ind * bps].reshape(n_ind, bps)
        codes = np.stack([(body >> (2 * k)) & 3 for k in range(4)], axis=2).reshape(n_ind, bps * 4)[:, :nsnp]
    g = recode[codes]
    chrs = np.array(chrs, dtype=np.int32)
    poss = np.array(poss) / 1000000  # PLINK_POS_SCALING
    keep = (chrs != 0) & (chrs < 23)
    return sort_snps(np.ascontiguousarray(g[:, keep]), poss[keep], chrs[keep])


def read_plink_raw(prefix):
    """PLINK input for relate_hip_load_bed: the .bed payload as it lies in the file (SNP-major only) plus the short
    per-SNP vectors after doBimFile's autosome filter and mysort -- the genotype matrix itself is never unpacked on
    the host.  Returns (payload uint8[n_snp * ceil(N/4)], N, cols, pos, chr)."""
    chrs, poss = read_bim(prefix + ".bim")
    with open(prefix + ".fam") as fh:
        n_ind = sum(1 for line in fh if line.strip())
    raw = np.fromfile(prefix + ".bed", dtype=np.uint8)
    if raw[0] != 0x6C or raw[1] != 0x1B:
        raise ValueError("Input file does not appear to be a .bed file")
    if not raw[2]:
        raise ValueError("individual-major .bed: use load_plink()")
    nsnp = len(chrs)
    bps = (n_ind + 3) // 4
    if raw.size < 3 + nsnp * bps:
        raise ValueError("Unexpected end of file reached")
    chrs = np.array(chrs, dtype=np.int32)
    poss = np.array(poss) / 1000000  # PLINK_POS_SCALING
    rows = np.nonzero((chrs != 0) & (chrs < 23))[0]
    order = np.lexsort((poss[rows], chrs[rows]))
    cols = rows[order].astype(np.int32)
    p = poss[cols]
    if len(p) > 1 and (p[1:] == p[:-1]).any():
        raise ValueError("Non unique position")
    return raw[3:3 + nsnp * bps], n_ind, cols, np.ascontiguousarray(p), np.ascontiguousarray(chrs[cols])


def keep_list_pairs(path):
    idx = [i for i, f in enumerate(read_vector(path, int)) if f == 1]
    return np.array([(a, b) for n, a in enumerate(idx) for b in idx[n + 1:]], dtype=np.int32).reshape(-1, 2)


def all_pairs(N):
    i, j = np.triu_indices(N, 1)
    return np.stack([i, j], axis=1).astype(np.int32)


def fmt_g(v, prec=4):
    """ostream << setprecision(prec) << double."""
    return "%.*g" % (prec, v)


def write_post_rows(fh, post, k1_only):
    """write_dMatrix_to_file for one pair: IBD0 row, then the IBD1 row unless k2 is fixed at 0; -1 marks dropped SNPs."""
    for row in ((2,) if k1_only else (2, 1)):
        fh.write("".join(("-1\t" if v == -1.0 else fmt_g(v) + "\t") for v in post[row]))
        fh.write("\n")


def write_post_block(fh, post, k1_only):
    """write_dMatrix_to_file for a decoded block (P x 3 x M) into a file opened in binary mode: the same bytes as
    write_post_rows pair by pair, formatted by the library's host threads (relate_format_post_rows; 10^8 values a second
    instead of 10^6)."""
    import ctypes as C
    from . import api
    L = api.load_library()
    post = np.ascontiguousarray(post, dtype=np.float64)
    P, three, M = post.shape
    assert three == 3
    stride = L.relate_post_text_stride(M)
    rpp = 1 if k1_only else 2
    group = max(1, min(P, (256 << 20) // (stride * rpp)))
    out = C.create_string_buffer(group * rpp * stride)
    lens = (C.c_size_t * (group * rpp))()
    view = memoryview(out)
    for g0 in range(0, P, group):
        ng = min(group, P - g0)
        rc = L.relate_format_post_rows(post[g0:g0 + ng].ctypes.data_as(api.dp), ng, M, int(k1_only), out, lens)
        if rc:
            raise api.RelateError(L.relate_hip_last_error(None).decode())
        for r in range(ng * rpp):
            fh.write(view[r * stride:r * stride + lens[r]])

