"""PLINK .bed ingestion on the GPU (relate_hip_load_bed) against the byte-matrix route (relate_hip_load) fed by the
host-side restatement of bed_to_iMatrix + doBimFile + mysort (relate_amd/io.py:load_plink;
src/filereader_and_conversions.cpp:483-657, src/asort.cpp:37-88).  Integer work: bit-exact.  Nothing here reads
/root/reference; data/500.* are the reference's own example files kept as fixtures."""
import os

import numpy as np
import pytest

import relate_amd
from relate_amd import io as rio

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")


def write_plink(prefix, codes, chrs, bp):
    """codes: n_snp x n_ind PLINK 2-bit codes."""
    n_snp, n_ind = codes.shape
    bps = (n_ind + 3) // 4
    pad = np.zeros((n_snp, bps * 4), dtype=np.uint8)
    pad[:, :n_ind] = codes
    body = (pad[:, 0::4] | (pad[:, 1::4] << 2) | (pad[:, 2::4] << 4) | (pad[:, 3::4] << 6)).astype(np.uint8)
    with open(prefix + ".bed", "wb") as fh:
        fh.write(bytes([0x6C, 0x1B, 0x01]))
        fh.write(body.tobytes())
    with open(prefix + ".bim", "w") as fh:
        for s in range(n_snp):
            fh.write("%d\trs%d\t0\t%d\tA\tG\n" % (chrs[s], s, bp[s]))
    with open(prefix + ".fam", "w") as fh:
        for i in range(n_ind):
            fh.write("F%d I%d 0 0 1 -9\n" % (i, i))


def synthetic(tmp_path, n_ind, n_snp, seed):
    rng = np.random.default_rng(seed)
    freq = rng.uniform(0.1, 0.9, size=n_snp)
    dose = rng.binomial(2, freq[:, None], size=(n_snp, n_ind))
    codes = np.array([0, 2, 3], dtype=np.uint8)[dose]          # hom A1, het, hom A2
    codes[rng.random(codes.shape) < 0.02] = 1                  # missing
    chrs = rng.choice([0, 1, 2, 3, 22, 23, 25], size=n_snp)    # 0 and >= 23 are dropped by doBimFile
    bp = rng.choice(np.arange(1, 50_000_000), size=n_snp, replace=False)
    prefix = str(tmp_path / ("syn%d" % seed))
    write_plink(prefix, codes, chrs, bp)
    return prefix


def test_raw_reader_matches_matrix_reader(tmp_path):
    """host side only: unpacking the payload rows that read_plink_raw selects gives load_plink's matrix."""
    for prefix in (os.path.join(DATA, "500"), synthetic(tmp_path, 37, 300, 5)):
        g, pos, chr_ = rio.load_plink(prefix)
        payload, n_ind, cols, pos2, chr2 = rio.read_plink_raw(prefix)
        bps = (n_ind + 3) // 4
        rows = payload.reshape(-1, bps)[cols]
        codes = np.stack([(rows >> (2 * k)) & 3 for k in range(4)], axis=2).reshape(len(cols), bps * 4)[:, :n_ind].T
        assert np.array_equal(np.array([3, 0, 2, 1], dtype=np.uint8)[codes], g)
        assert np.array_equal(pos, pos2) and np.array_equal(chr_, chr2)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["500", "syn37", "syn64", "syn130"])
def test_bed_route_equals_matrix_route(case, tmp_path):
    if case == "500":
        prefix = os.path.join(DATA, "500")
    else:
        prefix = synthetic(tmp_path, int(case[3:]), 400, 11)
    opt = rio.read_options(os.path.join(DATA, "options.all.pair.txt"))
    g, pos, chr_ = rio.load_plink(prefix)
    A = relate_amd.Relate(0)
    Ma = A.init_globals(g, pos, chr_, opt)
    payload, n_ind, cols, pos2, chr2 = rio.read_plink_raw(prefix)
    B = relate_amd.Relate(0)
    Mb = B.init_globals_bed(payload, n_ind, cols, pos2, chr2, opt)
    assert Ma == Mb and np.array_equal(A.keep, B.keep)
    ga, pa, ma = A.stripped()
    gb, pb, mb = B.stripped()
    assert np.array_equal(ga, gb)
    assert np.array_equal(pa, pb) and np.array_equal(ma, mb)
    ha, hb = A.hap_global(), B.hap_global()
    for k in ha:
        assert np.array_equal(ha[k], hb[k], equal_nan=k != "choose"), k
    pairs = [(0, 1), (2, n_ind - 1), (3, 7)]
    par = np.array([[(0.05, 0.1, 0.3), (0.02, 0.0, 0.5)]] * len(pairs))
    assert np.array_equal(A.like(pairs, par), B.like(pairs, par))


def test_read_bim_follows_the_reference_reader(tmp_path):
    """relate_amd.io.read_bim: six columns per line, reading stops at the first empty line (doBimFile), a line with
    another column count is an error."""
    import pytest
    from relate_amd import io as rio
    lines = ["%d\trs%d\t0\t%d\tA\tG" % (1 + n // 4, n, 1000 * n + 7) for n in range(12)]
    p = tmp_path / "a.bim"
    p.write_text("\n".join(lines) + "\n")
    c, q = rio.read_bim(str(p))
    assert c == [1 + n // 4 for n in range(12)] and q == [float(1000 * n + 7) for n in range(12)]
    p.write_text("\n".join(lines))                                   # no newline at the end
    assert rio.read_bim(str(p)) == (c, q)
    p.write_text("\n".join(lines[:5]) + "\n\n" + "\n".join(lines[5:]) + "\n")   # stops at the empty line
    assert rio.read_bim(str(p)) == (c[:5], q[:5])
    p.write_text("\n".join(lines[:5]) + "\n \t\n" + "\n".join(lines[5:]) + "\n")
    assert rio.read_bim(str(p)) == (c[:5], q[:5])
    p.write_text("\n".join(lines[:3]) + "\n1 rs 0 5 A\n")             # five columns
    with pytest.raises(ValueError):
        rio.read_bim(str(p))
    p.write_text("1 rs1 0 5 A G 2 rs2 0\n9 A G\n")                    # twelve tokens on two lines, not 6 + 6
    with pytest.raises(ValueError):
        rio.read_bim(str(p))
