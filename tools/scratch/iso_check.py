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
allp = synth.all_pairs(N)
edge = np.array([(0.05, 0.0, 0.0), (0.05, 1.0, 0.0), (0.05, 0.3, 0.0)])
for env, npairs, rep in (("1", 3, 1), ("100000000", 3, 1), (None, 900, 24)):
    if env is None:
        os.environ.pop("RELATE_SINGLE_SWEEP_LANES", None)
    else:
        os.environ["RELATE_SINGLE_SWEEP_LANES"] = env
    H = relate_amd.Relate(0)
    H.init_globals(g, pos, chr_, opt)
    eb = np.broadcast_to(np.tile(edge, (rep, 1)), (npairs, 3 * rep, 3)).copy()
    v = H.like(allp[:npairs], eb)
    print(env, H.last_chunks(), v[:3, :3].tolist(), flush=True)
