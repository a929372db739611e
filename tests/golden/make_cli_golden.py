#!/usr/bin/env python3
"""Generates tests/golden/cli/* by running the REFERENCE's own command-line program (oracle/_ref/relateHMM_ref:
built from /root/reference/src by `make -C oracle refcli`, with oracle/pin_time.c pinning srand(time(0)) to 4242)
on the fixture data.  Run here, in the build container; the outputs are committed data.

    c5_*  config 5: options.linkage.txt -file dat_small -d dat_small.keep   (190 pairs, pruned to 415 SNPs)
    c1_*  config 1: options.single.pair.txt -file dat_small -choose 0       (pair (0,1) through init_globals + sgl)
    c1b_* config 1 as BASELINE.json states it: the same command without -choose 0, i.e. the program's default route
          for a lone pair, single_pair / getPars (src/runRelate.cpp:419-420, src/relateHMM.cpp:3595-3737, 3049-3223):
          the fixture that pins "the legacy route equals init_globals + sgl on this data" 
    c2_*  config 2: options.all.pair.txt -file dat_small -d dat_small.keep  (README.md:71; 190 pairs, 2-parameter calcA
          mode, ~70 s on one core): stripped dumps, keep.index, the .k file, and of the 2.6 MB .post file only what is
          stable -- per pair the SNPs printed as -1 (c2_post_dropped.json); the reference's -logL per pair, which its
          program never prints, comes from the same sources through the harness with the same pinned seed
          (c2_ref_fit.json: a, k2, k1, -logL, timesRun, timesConverged)
    c3_*  config 3: options.all.pair.txt -plink 500                         (66 pairs; chaotic 2-parameter mode:
                                                                             kept for the stripped dumps and format)
    c6_*  synthetic PLINK files data/syn6.* (16 individuals x 1500 .bim lines in shuffled order, chromosomes 0, 23
          and 25 among them, 2 % missing; written here by tests/test_bed_ingest.py:synthetic, seed 6, and committed):
          options.all.pair.txt -plink syn6        (doBimFile filter, (chr,pos) sort, MAF filter: dumps; 120 pairs)
    c7_*  the same files, options.linkage.txt -plink syn6 -d syn6.keep (all 16 kept): pruning on top, 1-parameter fit
    c8_*  flat files data/syn8.geno/.pos/.chr (12 individuals x 800 SNPs of 5 chromosomes, columns in shuffled order,
          3 % missing; numpy default_rng(8), committed): options.linkage.txt -g -p -c -d syn8.keep  (mysort on the matrix)
"""
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DATA = os.path.join(HERE, "data")
REF = os.path.join(ROOT, "oracle", "_ref", "relateHMM_ref")
OUT = os.path.join(HERE, "cli")

RUNS = {
    "c5": ["-o", DATA + "/options.linkage.txt", "-file", DATA + "/dat_small", "-d", DATA + "/dat_small.keep",
           "-post", "out.post", "-k", "out.k"],
    "c1": ["-o", DATA + "/options.single.pair.txt", "-file", DATA + "/dat_small", "-choose", "0",
           "-post", "out.post", "-k", "out.k"],
    "c3": ["-o", DATA + "/options.all.pair.txt", "-plink", DATA + "/500", "-post", "out.post", "-k", "out.k"],
    "c1b": ["-o", DATA + "/options.single.pair.txt", "-file", DATA + "/dat_small", "-post", "out.post", "-k", "out.k"],
    "c2": ["-o", DATA + "/options.all.pair.txt", "-file", DATA + "/dat_small", "-d", DATA + "/dat_small.keep",
           "-post", "out.post", "-k", "out.k"],
}

if not os.path.exists(DATA + "/syn6.bed"):
    import pathlib
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_bed_ingest import synthetic
    with tempfile.TemporaryDirectory() as td:
        prefix = synthetic(pathlib.Path(td), 16, 1500, 6)
        for ext in (".bed", ".bim", ".fam"):
            shutil.copy(prefix + ext, DATA + "/syn6" + ext)
    with open(DATA + "/syn6.keep", "w") as fh:
        fh.write(" ".join(["1"] * 16) + "\n")
if not os.path.exists(DATA + "/syn8.geno"):
    import numpy as np
    rng = np.random.default_rng(8)
    n, m = 12, 800
    chrom = np.sort(rng.integers(1, 6, size=m))
    bp = np.concatenate([np.sort(rng.choice(np.arange(1, 40_000_000), size=int((chrom == c).sum()), replace=False))
                         for c in range(1, 6)])
    freq = rng.uniform(0.1, 0.9, size=m)
    hap = rng.random((2 * n, m)) < freq
    for s_ in range(1, m):                      # LD: copy the previous SNP's allele with probability 0.4
        cp = rng.random(2 * n) < 0.4
        hap[cp, s_] = hap[cp, s_ - 1]
    geno = 1 + hap[0::2].astype(int) + hap[1::2].astype(int)
    geno[rng.random(geno.shape) < 0.03] = 0
    order = rng.permutation(m)
    with open(DATA + "/syn8.geno", "w") as fh:
        for i in range(n):
            fh.write(" ".join(str(v) for v in geno[i, order]) + "\n")
    with open(DATA + "/syn8.pos", "w") as fh:
        fh.write(" ".join("%.6f" % (bp[c] / 1e6) for c in order) + "\n")
    with open(DATA + "/syn8.chr", "w") as fh:
        fh.write(" ".join(str(chrom[c]) for c in order) + "\n")
    with open(DATA + "/syn8.keep", "w") as fh:
        fh.write(" ".join(["1"] * n) + "\n")
RUNS["c8"] = ["-o", DATA + "/options.linkage.txt", "-g", DATA + "/syn8.geno", "-p", DATA + "/syn8.pos", "-c",
              DATA + "/syn8.chr", "-d", DATA + "/syn8.keep", "-post", "out.post", "-k", "out.k"]
RUNS["c6"] = ["-o", DATA + "/options.all.pair.txt", "-plink", DATA + "/syn6", "-post", "out.post", "-k", "out.k"]
RUNS["c7"] = ["-o", DATA + "/options.linkage.txt", "-plink", DATA + "/syn6", "-d", DATA + "/syn6.keep", "-post", "out.post",
              "-k", "out.k"]

os.makedirs(OUT, exist_ok=True)
for name, args in RUNS.items():
    with tempfile.TemporaryDirectory() as td:
        env = dict(os.environ, RELATE_PIN_TIME="4242")
        subprocess.run([REF] + args, cwd=td, env=env, check=True, stdout=subprocess.DEVNULL)
        for f in ("out.post", "out.k", "stripped.geno", "stripped.pos", "stripped.chr", "keep.index"):
            src = os.path.join(td, f)
            if name == "c2" and f == "out.post":  # keep the dropped-SNP pattern of every pair (rows IBD0, IBD1 per pair)
                import json
                rows = [ln.split("\t") for ln in open(src).read().splitlines()]
                drop = [[i for i, v in enumerate(r) if v.strip() == "-1"] for r in rows]
                assert all(a == b for a, b in zip(drop[0::2], drop[1::2])) and len(rows) == 380
                with open(os.path.join(OUT, "c2_post_dropped.json"), "w") as fh:
                    json.dump({"rows": len(rows), "cols": len([v for v in rows[0] if v.strip()]), "dropped": drop[0::2]}, fh)
                continue
            if name in ("c3", "c6", "c7", "c8") and f == "out.post":
                continue  # 0.5 MB of posteriors at chaotic estimates: not a usable pin
            if os.path.exists(src):
                shutil.copy(src, os.path.join(OUT, "%s_%s" % (name, f)))
# c2: the reference's full-precision fit per pair through the harness (same sources, same pinned seed)
import json  # noqa: E402
import sys  # noqa: E402
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathcheck as pc  # noqa: E402
import refload  # noqa: E402
ref = refload.Ref()
g, pos, chr_ = refload.load_dataset("dat_small")
opt = dict(pc.options("options.all.pair.txt"))
pairs = [tuple(int(v) for v in p) for p in refload.keep_pairs(os.path.join(DATA, "dat_small.keep"))]
ref.init(g, pos, chr_, opt)
try:
    fit = []
    for i, j in pairs:
        x = ref.sgl(i, j, seed=4242)
        fit.append({"pair": [i, j], "a": x["a"], "k2": x["k2"], "k1": x["k1"], "nll": x["nll"],
                    "timesRun": x["timesRun"], "timesConv": x["timesConv"]})
finally:
    ref.fini()
with open(os.path.join(OUT, "c2_ref_fit.json"), "w") as fh:
    json.dump({"options": "options.all.pair.txt", "seed": 4242, "pairs": fit}, fh)
print(sorted(os.listdir(OUT)))
