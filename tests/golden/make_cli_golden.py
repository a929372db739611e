#!/usr/bin/env python3
"""Generates tests/golden/cli/* by running the REFERENCE's own command-line program (oracle/_ref/relateHMM_ref:
built from /root/reference/src by `make -C oracle refcli`, with oracle/pin_time.c pinning srand(time(0)) to 4242)
on the fixture data.  Run here, in the build container; the outputs are committed data.

    c5_*  config 5: options.linkage.txt -file dat_small -d dat_small.keep   (190 pairs, pruned to 415 SNPs)
    c1_*  config 1: options.single.pair.txt -file dat_small -choose 0       (pair (0,1) through init_globals + sgl)
    c3_*  config 3: options.all.pair.txt -plink 500                         (66 pairs; chaotic 2-parameter mode:
                                                                             kept for the stripped dumps and format)
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

os.makedirs(OUT, exist_ok=True)
for name, args in RUNS.items():
    with tempfile.TemporaryDirectory() as td:
        env = dict(os.environ, RELATE_PIN_TIME="4242")
        subprocess.run([REF] + args, cwd=td, env=env, check=True, stdout=subprocess.DEVNULL)
        for f in ("out.post", "out.k", "stripped.geno", "stripped.pos", "stripped.chr", "keep.index"):
            src = os.path.join(td, f)
            if name == "c3" and f == "out.post":
                continue  # 0.5 MB of posteriors at chaotic estimates: not a usable pin
            if os.path.exists(src):
                shutil.copy(src, os.path.join(OUT, "%s_%s" % (name, f)))
print(sorted(os.listdir(OUT)))
