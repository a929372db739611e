#!/usr/bin/env python3
"""Generates tests/golden/hmmtest/golden.json by running the REFERENCE's linkage tester (oracle/_ref/HMMtest_ref:
src/HMMtest.txt compiled where it lies by `make -C oracle reftest`, srand(time(NULL)) pinned to 4242 by
oracle/pin_time.c) on the reference's own config-5 outputs kept under tests/golden/cli/.  Run here, in the build
container; the JSON is committed data."""
import json
import os
import re
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "HMMtest_ref")
POST = os.path.join(HERE, "cli", "c5_out.post")
POS = os.path.join(HERE, "cli", "c5_stripped.pos")
KF = os.path.join(HERE, "cli", "c5_out.k")
CC = os.path.join(HERE, "data", "dat_small.cc")
MIX = os.path.join(HERE, "hmmtest", "mixed.cc")  # this repo's own file: statuses interleaved, so mixed pairs exist

CASES = {
    "default": ["-P", POST, "-D", CC, "-infer", "1", "-p", POS, "-Nsim", "200"],
    "ccall2": ["-P", POST, "-D", CC, "-infer", "1", "-p", POS, "-Nsim", "100", "-ccAll", "2"],
    "missing": ["-P", POST, "-D", CC, "-infer", "0", "-Nsim", "100"],
    "kfile": ["-P", POST, "-D", CC, "-infer", "1", "-p", POS, "-Nsim", "100", "-K", KF],
    "mixed0": ["-P", POST, "-D", MIX, "-infer", "1", "-p", POS, "-Nsim", "100"],
    "mixed1": ["-P", POST, "-D", MIX, "-infer", "1", "-p", POS, "-Nsim", "100", "-ccAll", "1"],
    "mixed2": ["-P", POST, "-D", MIX, "-infer", "0", "-Nsim", "100", "-ccAll", "2"],
}

out = {"seed": 4242, "cases": {}}
for name, args in CASES.items():
    with tempfile.TemporaryDirectory() as td:
        env = dict(os.environ, RELATE_PIN_TIME="4242")
        r = subprocess.run([REF] + args + ["-c", "cov"], cwd=td, env=env, check=True, capture_output=True, text=True)
        stat = [float(v) for v in open(os.path.join(td, "stat.txt")).read().split()]
        xmax = [float(v) for v in open(os.path.join(td, "x.txt")).read().split()]
        covf = [f for f in os.listdir(td) if f.startswith("cov.")][0]
        cov = [[float(v) for v in line.split()] for line in open(os.path.join(td, covf)).read().splitlines()[:6]]
        m = re.search(r"pmaxsim:([-\d.eE+naif]+)", r.stdout)
        m2 = re.search(r"psim/nsim:([-\d.eE+naif]+)\s+psigsim:([-\d.eE+naif]+)", r.stdout)
        m3 = re.search(r"numCases=(\d+),\s*numControls:(\d+)", r.stdout)
        out["cases"][name] = dict(args=[a.replace(HERE + "/", "") for a in args], stat=stat, xmax=xmax,
                                  cov_first_rows=cov, pmaxsim=float(m.group(1)), psim_over_nsim=float(m2.group(1)),
                                  psigsim=float(m2.group(2)), n_case_pairs=int(m3.group(1)),
                                  n_control_pairs=int(m3.group(2)))
os.makedirs(os.path.join(HERE, "hmmtest"), exist_ok=True)
with open(os.path.join(HERE, "hmmtest", "golden.json"), "w") as fh:
    json.dump(out, fh)
print({k: (v["pmaxsim"], v["psim_over_nsim"], v["psigsim"], v["n_case_pairs"], v["n_control_pairs"]) for k, v in out["cases"].items()})
