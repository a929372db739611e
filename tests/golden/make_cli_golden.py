#!/usr/bin/env python3
"""Generates tests/golden/cli/* by running the REFERENCE's own command-line program (oracle/_ref/relateHMM_ref:
built from /root/reference/src by `make -C oracle refcli`, with oracle/pin_time.c pinning srand(time(0)) to 4242)
on the fixture data.  Run here, in the build container; the outputs are committed data.

    c5_*  config 5: options.linkage.txt -file dat_small -d dat_small.keep   (190 pairs, pruned to 415 SNPs)
    c1_*  config 1: options.single.pair.txt -file dat_small -choose 0       (pair (0,1) through init_globals + sgl)
    c3_*  config 3: options.all.pair.txt -plink 500                         (66 pairs; chaotic 2-parameter mode:
                                                                             kept for the stripped dumps and format)
    c6_*  synthetic PLINK files data/syn6.* (16 individuals x 1500 .bim lines in shuffled order, chromosomes 0, 23
          and 25 among them, 2 % missing; written here by tests/test_bed_ingest.py:synthetic, seed 6, and committed):
          options.all.pair.txt -plink syn6        (doBimFile filter, (chr,pos) sort, MAF filter: dumps; 120 pairs)
    c7_*  the same files, options.linkage.txt -plink syn6 -d syn6.keep (all 16 kept): pruning on top, 1-parameter fit
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
            if name in ("c3", "c6", "c7") and f == "out.post":
                continue  # 0.5 MB of posteriors at chaotic estimates: not a usable pin
            if os.path.exists(src):
                shutil.copy(src, os.path.join(OUT, "%s_%s" % (name, f)))
print(sorted(os.listdir(OUT)))
