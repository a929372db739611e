#!/usr/bin/env python3
"""One whole per-GPU shard of config 4 through the C++ host driver (runs ON THE GPU BOX): synthetic PLINK files of
N individuals x M SNPs (config 4: 2000 x 500000), all pairs, options.all.pair.txt semantics, rank 0 of 8 -- the
249 875 pairs one GPU of an 8-GPU node gets from relate_hip_shard -- fitted block by block with no .post output
(`relateHMM_hip -plink d -rank 0 -nranks 8 -gather none -est out.est`).  Records: wall-clock and pairs/s over the whole
shard, the per-block lines of RELATE_FIT_TRACE=1, the driver's peak resident set, the status histogram, and the
estimates of the planted related pairs that fall inside the shard against what was planted.
Usage: python tools/shard_run.py [N] [M] [NRANKS] [RANK] > profiles/r03_shard_run.txt"""
import os, resource, shutil, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from relate_amd import api, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
NR = int(sys.argv[3]) if len(sys.argv) > 3 else 8
RK = int(sys.argv[4]) if len(sys.argv) > 4 else 0
OPT = os.path.join(ROOT, "tests", "golden", "data", "options.all.pair.txt")
CLI = os.path.join(ROOT, "relate_amd", "relateHMM_hip")
t0 = time.perf_counter()
g, pos, chr_, related = synth.make_dataset(N, M, device="cuda:0")
g = np.asarray(g, dtype=np.uint8)
tmp = tempfile.mkdtemp(prefix="relate_shard_")
try:
    codes = np.array([1, 3, 2, 0], dtype=np.uint8)[g.T]          # inverse of bed_to_iMatrix's recode + swap
    bps = (N + 3) // 4
    pad = np.zeros((M, bps * 4), dtype=np.uint8)
    pad[:, :N] = codes
    body = (pad[:, 0::4] | (pad[:, 1::4] << 2) | (pad[:, 2::4] << 4) | (pad[:, 3::4] << 6)).astype(np.uint8)
    with open(os.path.join(tmp, "d.bed"), "wb") as fh:
        fh.write(bytes([0x6C, 0x1B, 0x01])); fh.write(body.tobytes())
    chr_ = np.asarray(chr_)
    bp = np.rint(np.asarray(pos) * 1e6).astype(np.int64)         # the data set's own positions (Mb) as base pairs
    for c in np.unique(chr_):                                     # strictly increasing inside a chromosome
        sel = np.where(chr_ == c)[0]
        bp[sel] = np.maximum.accumulate(bp[sel] - np.arange(len(sel))) + np.arange(len(sel))
        assert (np.diff(bp[sel]) > 0).all()
    with open(os.path.join(tmp, "d.bim"), "w") as fh:
        fh.write("".join("%d\trs%d\t0\t%d\tA\tG\n" % (chr_[s], s, bp[s]) for s in range(M)))
    with open(os.path.join(tmp, "d.fam"), "w") as fh:
        fh.write("".join("F%d I%d 0 0 1 -9\n" % (i, i) for i in range(N)))
    del g, codes, pad, body
    Pall = N * (N - 1) // 2
    b, e = api.shard(Pall, RK, NR)
    print("synthetic PLINK files: %d individuals x %d SNPs written in %.1f s; shard %d of %d = pairs %d .. %d (%d pairs)" % (
        N, M, time.perf_counter() - t0, RK, NR, b, e, e - b), flush=True)
    env = dict(os.environ, RELATE_FIT_TRACE="1")
    t0 = time.perf_counter()
    errf = os.path.join(tmp, "trace.err")
    with open(errf, "w") as eh, open(os.path.join(tmp, "trace.out"), "w") as oh:
        pr = subprocess.Popen([CLI, "-o", OPT, "-plink", "d", "-post", "", "-k", "out.k", "-est", "out.est", "-seed", "4242",
                               "-v", "0", "-rank", str(RK), "-nranks", str(NR), "-gather", "none", "-gpu", "0"], cwd=tmp,
                              env=env, stdout=oh, stderr=eh)
        while pr.poll() is None:       # a line a minute: the box takes a silent command for a hung one
            time.sleep(30)
            sys.stderr.write("shard run: %.0f s, %d fit blocks done\n" % (
                time.perf_counter() - t0, sum(l.startswith("fit block") for l in open(errf))))
            sys.stderr.flush()
    dt = time.perf_counter() - t0
    rss = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1048576.0

    class R:
        returncode, stdout, stderr = pr.returncode, open(os.path.join(tmp, "trace.out")).read(), open(errf).read()
    r = R
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    P = e - b
    print("relateHMM_hip -rank %d -nranks %d -gather none (no .post): %.1f s wall for %d pairs x %d SNPs = %.0f pairs/s = "
          "%.3e pair*SNP/s end to end; peak resident set of the driver %.2f GiB" % (RK, NR, dt, P, M, P / dt, P * M / dt, rss))
    blocks = [l for l in r.stderr.splitlines() if l.startswith("fit block")]
    print("fit blocks (RELATE_FIT_TRACE=1): %d" % len(blocks))
    for l in blocks:
        print("  " + l)
    # the likelihood launches of the whole run by slot bracket
    import re
    rows = []
    for l in r.stderr.splitlines():
        m = re.match(r"like launch: (\d+) lanes, (\d+) chunks, ([\d.]+) ms, (\d+) points", l)
        if m:
            rows.append((int(m.group(1)), int(m.group(2)), float(m.group(3)), int(m.group(4))))
    edges = [0, 45056, 65536, 98304, 131072, 196608, 262144, 393216, 10 ** 9]
    print("likelihood launches: %d, kernel time %.1f s; sweeps by slot bracket:" % (len(rows), sum(x[2] for x in rows) / 1e3))
    for lo, hi in zip(edges, edges[1:]):
        sel = [x for x in rows if lo < x[0] <= hi and x[1] == 1]
        if sel:
            ms = sum(x[2] for x in sel)
            print("  %7d..%-7d launches %5d  %9.1f ms  avg %6.1f ms  %6.0f G point-steps/s" % (
                lo, min(hi, 999999), len(sel), ms, ms / len(sel), sum(x[3] for x in sel) * M / ms / 1e6))
    sel = [x for x in rows if x[1] > 1]
    if sel:
        ms = sum(x[2] for x in sel)
        print("  cut along the SNP axis: launches %5d  %9.1f ms  %6.0f G point-steps/s" % (len(sel), ms, sum(x[3] for x in sel) * M / ms / 1e6))
    est = np.loadtxt(os.path.join(tmp, "out.est.rank%d" % RK))
    assert len(est) == P
    vals, cnt = np.unique(est[:, -1].astype(int), return_counts=True)
    print("status histogram: " + ", ".join("%d: %d" % (v, c) for v, c in zip(vals, cnt)))
    print("timesRun histogram: " + ", ".join("%d: %d" % (v, c) for v, c in zip(*np.unique(est[:, 6].astype(int), return_counts=True))))
    print("timesConverged histogram: " + ", ".join("%d: %d" % (v, c) for v, c in zip(*np.unique(est[:, 7].astype(int), return_counts=True))))
    k1, k2 = est[:, 4], est[:, 3]
    rel = {(i, j): (k, a) for i, j, k, a in related}
    isrel = np.array([(int(x[0]), int(x[1])) in rel for x in est])
    print("unrelated pairs in the shard: %d; k0 = 1 - k1 - k2: median %.4f, 1 %% quantile %.4f, pairs with k0 < 0.8: %d" % (
        int((~isrel).sum()), np.median(1 - k1[~isrel] - k2[~isrel]), np.quantile(1 - k1[~isrel] - k2[~isrel], 0.01),
        int(((1 - k1 - k2)[~isrel] < 0.8).sum())))
    print("planted related pairs in the shard: %d (planted k = (k0, k1, k2); estimates from the .est file)" % int(isrel.sum()))
    worst = 0.0
    for x in est[isrel]:
        k, a = rel[(int(x[0]), int(x[1]))]
        got = (1 - x[4] - x[3], x[4], x[3])
        worst = max(worst, max(abs(got[q] - k[q]) for q in range(3)))
        print("  pair (%d, %d): planted (%.2f, %.2f, %.2f)  estimated (%.3f, %.3f, %.3f)  a %.4f  -logL %.1f  runs %d" % (
            x[0], x[1], *k, *got, x[2], x[5], x[6]))
    print("largest |estimated - planted| over the planted pairs: %.3f" % worst)
    if NR > 1:
        print("PROJECTION (no multi-GPU run was made): %d ranks x this shard's %.0f s = the 8-GPU wall-clock of config 4 if every "
              "shard takes what rank %d's took; shards are static and equal in pair count (+-1)" % (NR, dt, RK))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
