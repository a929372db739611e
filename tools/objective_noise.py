#!/usr/bin/env python3
"""How far the HIP objective sits from the reference's like() in units of the last place: the device likelihood at the
end points of the reference's own runs (tests/golden/sensitivity_allpair.json, cli/c3_ref_fit.json: 8 per pair, whose
-logL the fixture holds to the last bit) against those values.  The result sizes the noise amplitude of the fixtures'
perturbed copies (tests/golden/make_sensitivity_allpair.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathcheck as pc  # noqa: E402
import relate_amd  # noqa: E402

H = relate_amd.Relate(0)
for name, f in (("dat_small", "sensitivity_allpair.json"), ("500", "cli/c3_ref_fit.json")):
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", f)))
    g, pos, chr_ = pc.dataset(name)
    opt = dict(pc.options(fx["options"]), seed=fx["seed"], seed_mode=0)
    H.init_globals(g, pos, chr_, opt)
    pairs = [tuple(p["pair"]) for p in fx["pairs"]]
    par = np.array([[r[:3] for r in p["runs"]] for p in fx["pairs"]])
    want = np.array([[r[3] for r in p["runs"]] for p in fx["pairs"]])
    got = H.like(pairs, par)
    ulp = (got - want) / np.spacing(want)
    q = np.quantile(np.abs(ulp), [0.5, 0.9, 0.99, 1.0])
    print("%s: %d points; (device - reference) in ulp: mean %.2f, rms %.2f, |.| median %.1f, 90 %% %.1f, 99 %% %.1f, max %.1f; "
          "equal to the bit: %d; max relative %.2e" % (name, ulp.size, ulp.mean(), np.sqrt((ulp ** 2).mean()), *q,
                                                      int((ulp == 0).sum()), np.abs(got / want - 1).max()))
