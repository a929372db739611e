"""Ad-hoc GPU check (not collected by pytest): prints parity numbers of every stage against the oracle and,
when oracle/_ref is present, against the compiled reference.  Run on the GPU box:
    python tests/adhoc_gpu_check.py > gpurun_out/adhoc.log
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracleload as O  # noqa: E402
import refload as R  # noqa: E402
import relate_amd as ra  # noqa: E402


def main():
    o = O.Oracle()
    try:
        ref = R.Ref()
    except Exception as e:  # noqa: BLE001
        print("no reference library:", e)
        ref = None
    g, pos, chr_ = R.load_dataset("dat_small")
    H = ra.Relate(0)
    for optf, pairs in (("options.single.pair.txt", [(0, 1), (5, 12), (2, 9), (3, 50), (100, 103)]),
                        ("options.all.pair.txt", [(0, 1), (2, 9), (7, 8)]),
                        ("options.linkage.txt", [(0, 1), (5, 12), (11, 19)])):
        opt = R.read_options(os.path.join(R.DATA, optf))
        t0 = time.time()
        M = H.init_globals(g, pos, chr_, opt)
        print("==", optf, "M", M, "precalc %.3fs" % (time.time() - t0))
        # oracle pre-calc
        g8 = g.astype(np.uint8)
        maf0 = o.maf(g8)
        keep = o.maf_keep(maf0, opt["min"])
        gs = g8[:, keep == 1]
        ps = pos[keep == 1]
        if opt["doPrune"]:
            k2 = o.prune(gs, opt["LD"], opt["back"], opt["prune_val"])
            full = keep.copy()
            full[np.where(keep == 1)[0][k2 == 0]] = 0
            keep = full
            gs = g8[:, keep == 1]
            ps = pos[keep == 1]
        print(" keep equal:", np.array_equal(keep, H.keep), int(keep.sum()))
        mafs = o.maf(gs)
        gg, gpos, gmaf = H.stripped()
        print(" stripped geno equal:", np.array_equal(gg, gs), " pos:", np.array_equal(gpos, ps), " maf:",
              np.array_equal(gmaf, mafs))
        hap = o.hap_global(gs, opt["back2"], opt["LD"])
        gh = H.hap_global()
        for k in ("mea", "pBA", "pBa", "pbA", "pba"):
            d = np.abs(gh[k] - hap[k])
            print("  hap %s bit-equal %s maxabs %.3g nmismatch %d" % (k, np.array_equal(gh[k], hap[k]), np.nanmax(d),
                                                                     int((d > 0).sum())))
        for (i, j) in pairs:
            st = o.pair_stage(gs, i, j, mafs, ps, opt["epsilon"], 1, opt["back"], opt["back2"], hap)
            kp, tt, S = H.pair_emissions(i, j)
            idx = np.where(kp == 1)[0]
            print(" pair", (i, j), "m", st["m"], "keep eq", np.array_equal(kp, st["keep"]), "t eq",
                  np.array_equal(tt[idx][1:], st["t"][1:st["m"]]), "S maxrel %.3g" % np.abs(S[:, idx] / st["S"] - 1).max())
            Sk = o.logs(st["S"])
            grid = [(0.05, 0.1, 0.3), (0.001, 0, 0), (0.001, 0, 1), (0.1, 0.5, 0.5), (0.149, 0.0, 0.25), (0.2, 0.1, 0.1),
                    (0.1, 0.7, 0.4), (0.0905, 0.0, 0.1124), (0.15, 0.25, 0.5), (0.01, 1.0, 0.0), (0.03, 0.0, 0.999)]
            got = H.like([(i, j)], np.array(grid)[None])[0]
            want = np.array([o.like(d, *Sk, st["t"], 0, opt["alim1"]) for d in grid])
            print("   like maxrel %.3g" % np.abs(got / want - 1).max(), got[:3], want[:3])
            for par in ((0.05, 0.2, 0.3), (0.0905, 0.0, 0.1124)):
                post, path, ns = H.decode([(i, j)], [par], want_path=True)
                po = o.decode(par, *Sk, st["t"], 0)
                vo = o.viterbi(par, *Sk, st["t"], 0)
                print("   decode", par, "maxabs %.3g" % np.abs(post[0][:, idx] - po).max(), "dropped -1:",
                      bool((post[0][:, kp == 0] == -1).all()), "nsnp", ns[0], "viterbi eq", np.array_equal(path[0][idx], vo),
                      int((path[0][idx] != vo).sum()))
        # full fit vs the compiled reference with pinned seed
        if ref is not None:
            ref.init(g, pos, chr_, opt)
            o2 = dict(opt, seed=4242, seed_mode=0)
            H.init_globals(g, pos, chr_, o2)
            t0 = time.time()
            res, post, path = H.sgl(pairs, want_path=True)
            print(" GPU sgl for %d pairs: %.3fs" % (len(pairs), time.time() - t0), H.timings())
            for (i, j), r in zip(pairs, res):
                x = ref.sgl(i, j, seed=4242)
                print("  fit", (i, j), "ref a %.10g k2 %.10g k1 %.10g nll %.12g runs %d/%d" % (
                    x["a"], x["k2"], x["k1"], x["nll"], x["timesConv"], x["timesRun"]))
                print("      ", "gpu a %.10g k2 %.10g k1 %.10g nll %.12g runs %d/%d u %.12g po %.12g st %d" % (
                    r["a"], r["k2"], r["k1"], r["negloglik"], r["timesConverged"], r["timesRun"], r["uLike"],
                    r["poLike"], r["status"]), " ref u %.12g po %.12g" % (x["uLike"], x["poLike"]))
            ref.fini()


if __name__ == "__main__":
    main()
