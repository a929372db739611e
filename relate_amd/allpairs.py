"""Multi-GPU all-pairs driver: one process per GPU (torch.distributed.run), pair blocks sharded across ranks, no
exchange while pairs are processed, one all_gather of the per-pair estimates at the end (RCCL over xGMI with the
"nccl" backend).  Same inputs and option file as the reference CLI / relate_amd/relateHMM_hip.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        -m relate_amd.allpairs -o options.all.pair.txt -file data/dat_small [-d keep] -out run1 [-seed N] [-no-post]

Outputs: every rank writes the posterior rows of its shard to <out>.post.rank<r> (in pair order, the reference's
.post format: concatenating the shards in rank order gives the single-process .post); rank 0 writes <out>.k (the
reference's format) and <out>.est (pair, a, k2, k1, -logL, runs, converged, nsnp, status at full precision) for
ALL pairs after the gather.  Posteriors are never gathered: at config 4 they are terabytes.
-post-format bin writes <out>.postbin.rank<r> instead: the same rows as raw little-endian float32, M values per row
(-1 = SNP dropped for the pair), with <out>.postbin.rank<r>.json = {"rows", "snps", "rows_per_pair", "first_pair"} --
the shard format for data sets whose text .post would not be practical (4 significant digits fit a float32).

Work is cut twice: the fit runs on as many pairs at a time as device memory holds records for (the library cuts a
call of -block pairs itself; with tens of thousands of lanes in flight the likelihood launches are whole turns of
single sweeps rather than chunked ones), and the posteriors of a fitted block are decoded and written in sub-blocks
bounded by a 1 GiB host buffer.

SURVEY.md section 8(e); pair order: fast_all_pairs / joblist_fast_all_pairs (src/relateHMM.cpp:3764-3765, 3851).
"""
import argparse
import os
import sys
import time

import numpy as np

from . import io as rio


def parse_args(argv=None):
    ap = argparse.ArgumentParser(prog="relate_amd.allpairs", prefix_chars="-")
    ap.add_argument("-o", dest="options", required=True)
    ap.add_argument("-file", dest="prefix")
    ap.add_argument("-g", dest="geno")
    ap.add_argument("-p", dest="pos")
    ap.add_argument("-c", dest="chr")
    ap.add_argument("-plink", dest="plink")
    ap.add_argument("-d", dest="keep")
    ap.add_argument("-out", dest="out", default="relate_out")
    ap.add_argument("-seed", dest="seed", type=int, default=None)
    ap.add_argument("-no-post", dest="no_post", action="store_true")
    ap.add_argument("-block", dest="block", type=int, default=0,
                    help="pairs per library call (0 = 65536; the library cuts a call into the blocks its memory holds)")
    ap.add_argument("-post-block", dest="post_block", type=int, default=0,
                    help="pairs per posterior decode call (0 = what a 1 GiB host buffer holds)")
    ap.add_argument("-post-format", dest="post_format", choices=("text", "bin"), default="text")
    return ap.parse_args(argv)


def load_inputs(a, raw_bed=False):
    """(geno N x M, pos, chr) -- or, for a SNP-major .bed when the handle can recode it on the GPU (raw_bed), the
    file's payload as it lies: ("bed", payload, N, cols, pos, chr); the genotype matrix is then never unpacked on the host."""
    if a.plink:
        front = a.plink[:-4] if a.plink.endswith((".bed", ".bim", ".fam")) else a.plink
        if raw_bed:
            with open(front + ".bed", "rb") as fh:
                head = fh.read(3)
            if len(head) == 3 and head[2]:
                return ("bed",) + tuple(rio.read_plink_raw(front))
        return rio.load_plink(front)
    if a.prefix:
        front = a.prefix
        for ext in (".geno", ".pos", ".chr"):
            if front.endswith(ext):
                front = front[:-len(ext)]
        return rio.load_text(front + ".geno", front + ".pos", front + ".chr")
    if not a.geno or not a.pos:
        raise SystemExit("You must supply atleast a genotype file -g and a positionfile -p (or -file / -plink)")
    return rio.load_text(a.geno, a.pos, a.chr)


def run(a, handle_factory, dist=None, coll_device=None):
    """Body of one rank.  handle_factory() -> object with init_globals / sgl (relate_amd.api.Relate, or a stand-in in
    the CPU tests).  dist: an initialised torch.distributed module or None for a single process."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    opt = rio.read_options(a.options)
    opt["seed_mode"] = 0
    opt["seed"] = int(a.seed if a.seed is not None else time.time()) & 0xFFFFFFFF
    marks = [("start", time.perf_counter())]
    H = handle_factory()
    inp = load_inputs(a, raw_bed=hasattr(H, "init_globals_bed"))
    is_bed = isinstance(inp[0], str)
    n_ind = inp[2] if is_bed else inp[0].shape[0]
    pairs = rio.keep_list_pairs(a.keep) if a.keep else rio.all_pairs(n_ind)
    marks.append(("read inputs", time.perf_counter()))
    from . import api, dist as rdist
    b, e = api.shard(len(pairs), rank, world)
    mine = pairs[b:e]
    if is_bed:
        M = H.init_globals_bed(inp[1], inp[2], inp[3], inp[4], inp[5], opt)
    else:
        M = H.init_globals(inp[0], inp[1], inp[2], opt)
    marks.append(("load + pre-calculation", time.perf_counter()))
    k1_only = opt["fixK"] == 1 and opt["fixK_val"] == 0
    block = a.block if a.block > 0 else 65536
    post_block = getattr(a, "post_block", 0) or max(1, min(block, (1 << 30) // (24 * M)))
    fits = []
    binary = getattr(a, "post_format", "text") == "bin"
    post_path = "%s.%s.rank%d" % (a.out, "postbin" if binary else "post", rank)
    fh = None if a.no_post else open(post_path, "wb")
    nrows = 0
    pin = None   # page-locked posterior buffer of one decode sub-block, reused
    try:
        rows = 1 if k1_only else 2
        for b0 in range(0, len(mine), block):
            blk = mine[b0:b0 + block]
            if fh is None:
                fits.extend(H.sgl(blk, want_post=False)[0])
                continue
            if len(blk) <= post_block and not binary:
                res, post, _ = H.sgl(blk)
                fits.extend(res)
                rio.write_post_block(fh, post, k1_only)
                nrows += len(blk) * rows
                continue
            res, _, _ = H.sgl(blk, want_post=False)   # fit (+ SNP counts); posteriors follow in sub-blocks
            fits.extend(res)
            for c0 in range(0, len(blk), post_block):
                sub = blk[c0:c0 + post_block]
                par = [(r["a"], r["k2"], r["k1"]) for r in res[c0:c0 + post_block]]
                if binary:   # float32 rows in writer order straight from the GPU: the shard's own layout
                    if pin is None and hasattr(H, "post_buffer"):
                        pin = H.post_buffer(min(post_block, len(blk)), rows, np.float32)
                    post, _ = H.decode_f32(sub, par, rows) if pin is None else H.decode_f32(sub, par, rows, out=pin)
                    fh.write(memoryview(np.ascontiguousarray(post, dtype="<f4")).cast("B"))
                else:
                    if pin is None and hasattr(H, "post_buffer"):
                        pin = H.post_buffer(min(post_block, len(blk)))
                    post, _, _ = H.decode(sub, par) if pin is None else H.decode(sub, par, out=pin)
                    rio.write_post_block(fh, post, k1_only)
                nrows += len(sub) * rows
    finally:
        if fh is not None:
            fh.close()
    if binary and not a.no_post:
        import json
        with open(post_path + ".json", "w") as fj:
            json.dump({"rows": nrows, "snps": int(M), "rows_per_pair": 1 if k1_only else 2, "first_pair": int(b),
                       "dtype": "<f4", "missing": -1}, fj)
    marks.append(("fit, decode, posterior rows", time.perf_counter()))
    local = rdist.fits_to_array(fits)
    full = rdist.gather_rows(local, len(pairs), device=coll_device) if dist is not None else local
    if rank == 0:
        F = {k: i for i, k in enumerate(rdist.FIT_FIELDS)}
        with open(a.out + ".k", "w") as fk, open(a.out + ".est", "w") as fe:
            for n in range(len(pairs)):
                r = full[n]
                fk.write(rio.fmt_g(r[F["k1"]]) + "\n")
                if not k1_only:
                    fk.write(rio.fmt_g(r[F["k2"]]) + "\n")
                fe.write("%d\t%d\t%.17g\t%.17g\t%.17g\t%.17g\t%d\t%d\t%d\t%d\n" % (
                    pairs[n][0], pairs[n][1], r[F["a"]], r[F["k2"]], r[F["k1"]], r[F["negloglik"]],
                    int(r[F["timesRun"]]), int(r[F["timesConverged"]]), int(r[F["nsnp"]]), int(r[F["status"]])))
    marks.append(("gather, .k / .est", time.perf_counter()))
    if rank == 0 and os.environ.get("RELATE_TIMINGS"):
        sys.stderr.write("timings: " + ", ".join("%s %.2f s" % (n, t - marks[i][1]) for i, (n, t) in enumerate(marks[1:])) + "\n")
    return full


def main(argv=None):
    a = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal switches for a one-GPU box, as in bench.py
    if os.environ.get("RELATE_BENCH_ONE_GPU") == "1":
        local = 0
    backend = os.environ.get("RELATE_BENCH_BACKEND", "nccl")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("relate_amd.allpairs needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    from . import api
    run(a, lambda: api.Relate(local), dist, torch.device("cuda", local) if backend == "nccl" else torch.device("cpu"))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
