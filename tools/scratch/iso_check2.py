import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathcheck as pc
import relate_amd
from relate_amd import synth
N, M = 200, 120000
g, pos, chr_, _ = synth.make_dataset(N, M, n_related=20)
opt = dict(pc.options("options.all.pair.txt"))
edge = np.array([[(0.05, 0.0, 0.0), (0.05, 1.0, 0.0), (0.05, 0.3, 0.0)]])
pair = [(0, 1)]
os.environ["RELATE_SINGLE_SWEEP_LANES"] = "1"
H1 = relate_amd.Relate(0)
os.environ["RELATE_SINGLE_SWEEP_LANES"] = "100000000"
H2 = relate_amd.Relate(0)
lo, hi = 600, M
def diff(m):
    out = []
    for H in (H1, H2):
        H.init_globals(np.ascontiguousarray(g[:, :m]), pos[:m], chr_[:m], opt)
        out.append(H.like(pair, edge)[0])
    return out, H2.last_chunks()
for m in (1200, 2400, 5000, 10000, 20000, 40000, 80000, 120000):
    (a, b), ch = diff(m)
    print(m, ch, a.tolist(), b.tolist(), flush=True)
