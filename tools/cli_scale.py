#!/usr/bin/env python3
"""End-to-end wall-clock of the host driver at a size where every stage matters: synthetic PLINK files (N individuals x
M SNPs, statistics of config 4), all pairs, options.all.pair.txt semantics, .post/.k written to a scratch directory.
Usage: python tools/cli_scale.py [N] [M] [KEEP]   (KEEP: only the pairs among the first KEEP individuals, through
relate_amd.allpairs -d keep -no-post -- the full-length run of a slice of config 4)"""
import os, shutil, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from relate_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
KEEP = int(sys.argv[3]) if len(sys.argv) > 3 else 0
OPT = os.path.join(ROOT, "tests", "golden", "data", "options.all.pair.txt")
CLI = os.environ.get("RELATE_CLI", os.path.join(ROOT, "relate_amd", "relateHMM_hip"))  # e.g. a sanitizer build
g, pos, chr_, _ = synth.make_dataset(N, M, n_related=8)
g = np.asarray(g, dtype=np.uint8)
tmp = tempfile.mkdtemp(prefix="relate_scale_")
try:
    codes = np.array([1, 3, 2, 0], dtype=np.uint8)[g.T]          # inverse of bed_to_iMatrix's recode + swap
    bps = (N + 3) // 4
    pad = np.zeros((M, bps * 4), dtype=np.uint8)
    pad[:, :N] = codes
    body = (pad[:, 0::4] | (pad[:, 1::4] << 2) | (pad[:, 2::4] << 4) | (pad[:, 3::4] << 6)).astype(np.uint8)
    with open(os.path.join(tmp, "d.bed"), "wb") as fh:
        fh.write(bytes([0x6C, 0x1B, 0x01])); fh.write(body.tobytes())
    chr_ = np.asarray(chr_)
    idx = np.arange(M) - np.searchsorted(chr_, chr_)              # index within the chromosome (chr_ is sorted)
    with open(os.path.join(tmp, "d.bim"), "w") as fh:
        fh.write("".join("%d\trs%d\t0\t%d\tA\tG\n" % (chr_[s], s, 1000 * idx[s] + 1) for s in range(M)))
    with open(os.path.join(tmp, "d.fam"), "w") as fh:
        fh.write("".join("F%d I%d 0 0 1 -9\n" % (i, i) for i in range(N)))
    P = N * (N - 1) // 2
    if KEEP:
        with open(os.path.join(tmp, "d.keep"), "w") as fh:
            fh.write(" ".join("1" if i < KEEP else "0" for i in range(N)) + "\n")
        P = KEEP * (KEEP - 1) // 2
    for label, extra in (() if KEEP else (("with .post text", ["-post", "out.post"]),)):
        t0 = time.perf_counter()
        r = subprocess.run([CLI, "-o", OPT, "-plink", "d", "-k", "out.k", "-est", "out.est", "-seed", "4242", "-v", "0"]
                           + extra, cwd=tmp, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        size = os.path.getsize(os.path.join(tmp, "out.post")) if os.path.exists(os.path.join(tmp, "out.post")) else 0
        print("relateHMM_hip %s: N=%d M=%d P=%d pairs: %.1f s wall = %.0f pairs/s; .post %.2f GB" % (
            label, N, M, P, dt, P / dt, size / 1e9), flush=True)
    env = dict(os.environ, PYTHONPATH=ROOT, RELATE_TIMINGS="1")
    legs = (("-no-post", ["-no-post"]), ("-post-format bin", ["-post-format", "bin"]))
    if KEEP:
        legs = (("-d keep -no-post", ["-d", "d.keep", "-no-post"]),)
    for label, extra in (() if "RELATE_CLI" in os.environ else legs):
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-m", "relate_amd.allpairs", "-o", OPT, "-plink", "d", "-out", "ap" + label[1:3],
                            "-seed", "4242"] + extra, cwd=tmp, env=env, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        print("relate_amd.allpairs %s: N=%d M=%d P=%d pairs: %.1f s wall = %.0f pairs/s" % (label, N, M, P, dt, P / dt),
              flush=True)
        if os.environ.get("RELATE_FIT_TRACE"):
            print("".join(l + "\n" for l in r.stderr.splitlines() if l.startswith(("fit block", "timings:"))), end="", flush=True)
            print(r.stdout[-1500:], flush=True)
        est = np.loadtxt(os.path.join(tmp, "ap" + label[1:3] + ".est"))
        print("  .est rows %d, status != 0: %d, k1 > 0.3: %d, mean -logL %.1f" % (
            len(est), int((est[:, -1] != 0).sum()), int((est[:, 4] > 0.3).sum()), est[:, 5].mean()), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
