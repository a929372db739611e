#!/usr/bin/env python3
"""bench.py -- all-pairs IBD HMM hot path on MI355X: pair x SNP HMM steps per second.

Workload (config.workload): the scale configuration of BASELINE.json (config 4: synthetic, 2,000 individuals x
500,000 SNPs, all-pairs, options.all.pair.txt semantics) processed in pair blocks.  One bench *step* = one pass
of the hot path over one block of P pairs with the genotype matrix and all per-SNP tables resident in HBM:
    K4 stage_pairs  (per-pair emissions, front half of sgl)
    K5 like         (forward likelihood at E parameter points per pair = what one BFGS iterate requests)
    K6 decode       (forward + backward sweep, the two posterior rows the .post writer takes written to HBM)
One HMM step = one update of the 3-state recursion for one (pair, SNP, evaluation point); a bench step therefore
counts P * M * (E + 2) HMM steps (E forward evaluations, decode forward, decode backward) -- `value` is that blend,
kept for continuity; SURVEY.md 8(d)'s three rates stand beside it at the top level of the line: like_point_steps_per_s
(K5), decode_pair_snp_per_s (K6, forward + backward = one step) and end_to_end_pair_snp_per_s (the fit leg: what a
user waits for).  Every rank owns a contiguous shard of the pair list (relate_hip_shard) and walks it block by block:
weak scaling, no data-path collective; the path's one exchange, the all-gather of the per-pair result records, goes
through the C ABI (relate_hip_gather over RCCL) and the line reports how many ranks its communicator held.

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (about 6.3 TB/s achievable)

# Algorithmic bytes per (pair, SNP) of one launch by this design: emission record 16 B (two
# planes: S1/S0, S2/S0; the distance of a step is the shared per-SNP one, M x 8 B for the whole block); packed genotypes
# 2 x 2 bit; tile checkpoint 84 B per 16 SNPs; posterior ROWS x 8 B (the rows the .post writer takes: IBD0 and IBD1 in
# this mode, src/relateHMM.cpp:3771-3782).  The decode is two sweeps over the records: chunk products + tile
# checkpoints, then the backward sweep that writes the posteriors.
CKPT = 84.0 / 16
REC = 16.0
ROWS = 2


def algorithmic_bytes(like_chunks, decode_chunks):
    return {"stage": REC + 0.5 + 0.25,
            "like": REC,
            "decode_fwd": REC + CKPT,
            "decode_bwd": REC + CKPT + 8.0 * ROWS}


# SURVEY.md section 8(d)'s figures, which price the path without materialised emissions: two 2-bit genotypes per forward
# step (K5), plus the posterior rows for the decode (K6: 0.5 + 8 x rows).
SURVEY_BYTES = {"like": 0.5, "decode": 0.5 + 8.0 * ROWS}


# Vector instructions per point-step of the likelihood kernels, counted by rocprofv3 (SQ_INSTS_VALU x 64 / point-steps;
# profiles/r02_pmc_like_sweeps.txt, profiles/r02_pmc_sq.json): sweeps 26.3 (single, two points per lane) to 28.9 (two-sided,
# one point per lane) -- 27.3, the two-sided two-points-per-lane figure, is used for a fit's mix; chunk-product sweep
# (three row vectors) 53.5 at this data set's SNP density (four fifths of the record groups take the shortest series).
# Peak = one FP64 instruction per lane per 4 cycles per SIMD at the 2.4 GHz boost clock; under this load the part holds
# 1.6-1.7 GHz and a pure FMA stream measures 28.8e12 lane-FMA/s (tools/ubench/fp64_ops.hip).
VALU_PER_STEP = {1: 27.3, 3: 53.5}
FP64_LANE_OPS_PEAK = 256 * 4 * 16 * 2.4e9  # 39.3e12 at the 2.4 GHz boost clock


def pmc_traffic(P, M, E):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs,
    corrected as MI355X_MICROARCH.md prescribes), valid only for the configuration they were taken on."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            t = json.load(fh)
    except OSError:
        return {}
    if (t.get("P"), t.get("M"), t.get("E")) != (P, M, E):
        return {}
    return {k: v["hbm_bytes"] for k, v in t["kernels"].items()}


def opts_all_pair():
    """data/options.all.pair.txt after the reference's float parse (src/filereader_and_conversions.cpp:65-91)."""
    f = lambda x: float(np.float32(x))  # noqa: E731
    return dict(min=f(0.01), epsilon=f(0.01), prune_val=f(0.2), alim0=f(0.001), alim1=f(1.0), fixA_val=0.0,
                fixK_val=0.0, phi=f(0.013), convTol=f(0.1), par=(f(0.3), 0.25, 0.5), back=5, doPrune=0, LD=1, ld_adj=1,
                double_recom=0, fixA=0, fixK=0, calcA=1, timesToConverge=3, timesToRun=8, doPar=0, seed_mode=0,
                seed=20240601)


def eval_points(P, E, seed=7):
    """E points per pair shaped like a calcA request: a centre (k2,k1) and central-difference neighbours."""
    rng = np.random.default_rng(seed)
    k2 = rng.uniform(0.02, 0.3, size=P)
    k1 = rng.uniform(0.1, 0.6, size=P)
    a = rng.uniform(0.02, 0.12, size=P)
    par = np.empty((P, E, 3))
    for e in range(E):
        d = (e - E // 2) * 4.4e-6
        par[:, e, 0] = a
        par[:, e, 1] = k2 + (d if e % 2 else 0)
        par[:, e, 2] = k1 + (0 if e % 2 else d)
    return par


def cpu_baseline(H, opt, E, want_seconds=12.0):
    """Same hot path on the host cores for a bounded sample of the same workload (rank 0, N=1 only)."""
    import oracleload
    o = oracleload.Oracle()
    ref = None
    try:
        import refload
        ref = refload.Ref()
    except Exception:  # noqa: BLE001
        ref = None
    gs, ps, mafs = H.stripped()
    Ms = min(H.M, 20000)
    gs = np.ascontiguousarray(gs[:, :Ms])
    ps, mafs = ps[:Ms].copy(), mafs[:Ms].copy()
    hap = o.hap_global(gs, opt["back2"], opt["LD"])  # pre-calc, not timed (the GPU's is not timed either)
    max_pairs = 400
    par = eval_points(max_pairs, E)
    steps = 0
    t0 = time.perf_counter()
    npairs = 0
    for p in range(max_pairs):
        i, j = 2 * p + 200, 2 * p + 201
        st = o.pair_stage(gs, i, j, mafs, ps, opt["epsilon"], 1, opt["back"], opt["back2"], hap)
        Sk = o.logs(st["S"])
        for e in range(E):
            if ref is not None:
                ref.like(par[p, e], *Sk, st["t"], 0, [opt["alim0"], opt["alim1"]])
            else:
                o.like(par[p, e], *Sk, st["t"], 0, opt["alim1"])
        if ref is not None:
            ref.decode(par[p, 0], *Sk, st["t"], 0)
        else:
            o.decode(par[p, 0], *Sk, st["t"], 0)
        steps += st["m"] * (E + 2)
        npairs += 1
        if time.perf_counter() - t0 > want_seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "pair*SNP HMM steps/s", "cores": 1,
            "kind": "reference" if ref is not None else "port",
            "sample": "%d pairs x first %d SNPs of the same synthetic matrix, E=%d; like/decode by %s, per-pair "
                      "emissions by the C port; %.1f s on one core" % (
                          npairs, Ms, E, "the compiled reference (oracle/_ref)" if ref is not None else "the C port",
                          dt)}


def cpu_baseline_threads(H, opt, E, want_seconds=6.0):
    """The C port (oracle/: the reference's algorithm restated, re-entrant where the reference keeps its state in
    statics) on the same bounded sample with one thread per host core of the box's share (pairs are independent):
    the all-cores companion of cpu_baseline.  ctypes releases the GIL around every call."""
    import threading
    import oracleload
    o = oracleload.Oracle()
    nt = max(1, min(16, os.cpu_count() or 1))
    gs, ps, mafs = H.stripped()
    Ms = min(H.M, 20000)
    gs = np.ascontiguousarray(gs[:, :Ms])
    ps, mafs = ps[:Ms].copy(), mafs[:Ms].copy()
    hap = o.hap_global(gs, opt["back2"], opt["LD"])
    max_pairs = 880  # individuals 200 .. 1959 in disjoint pairs; ~4 s on 16 threads
    par = eval_points(max_pairs, E)
    steps = [0] * nt
    done = [0] * nt
    t0 = time.perf_counter()

    def work(k):
        for p in range(k, max_pairs, nt):
            i, j = 2 * p + 200, 2 * p + 201
            st = o.pair_stage(gs, i, j, mafs, ps, opt["epsilon"], 1, opt["back"], opt["back2"], hap)
            Sk = o.logs(st["S"])
            for e in range(E):
                o.like(par[p, e], *Sk, st["t"], 0, opt["alim1"])
            o.decode(par[p, 0], *Sk, st["t"], 0)
            steps[k] += st["m"] * (E + 2)
            done[k] += 1
            if time.perf_counter() - t0 > want_seconds:
                break

    th = [threading.Thread(target=work, args=(k,)) for k in range(nt)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": sum(steps) / dt, "unit": "pair*SNP HMM steps/s", "cores": nt, "kind": "port",
            "sample": "%d pairs x first %d SNPs of the same synthetic matrix, E=%d; the C port on %d threads, %.1f s" % (
                sum(done), Ms, E, nt, dt)}


def fit_leg(H, allp, npairs, M, first=0, note=None):
    """Outside the timed region and not part of `value`: the whole estimation (relate_hip_fit: up to 8 restarts of the
    2-parameter L-BFGS-B search per pair, every objective value a likelihood sweep on the GPU) on the first pairs of the
    same data set, wall-clock around the C-ABI call -- what a user of the all-pairs scan waits for."""
    blk = allp[first:first + npairs]
    # The record buffer grows to this block size once per process (a scan fits hundreds of blocks with it): grow it on
    # another block of the same size first and report that time on its own.
    t0 = time.perf_counter()
    if len(allp) >= first + 2 * npairs:
        H.bench_like(allp[first + npairs:first + 2 * npairs], eval_points(npairs, 1), 1)
    t_grow = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = H.fit(blk)
    dt = time.perf_counter() - t0
    tm = H.timings()
    st = H.like_stats()
    # point-steps = (pair, SNP, parameter point) updates of all likelihood launches of the fit, counted by the library;
    # FP64 instructions per point-step by route from the PMC passes (VALU_PER_STEP); fraction of the 39.3e12 nominal
    # lane-instructions/s, over the wall-clock of the whole call (host optimiser steps and transfers included)
    steps = st["sweep_point_steps"] + st["chunked_point_steps"]
    insts = st["sweep_point_steps"] * VALU_PER_STEP[1] + st["chunked_point_steps"] * VALU_PER_STEP[3]
    return {"pairs": int(npairs), "seconds": dt, "buffer_growth_seconds": t_grow, "pairs_per_s": npairs / dt, "pair_snp_per_s": npairs * M / dt,
            "like_kernel_seconds": tm["like_ms"] / 1e3, "restarts_per_pair": sum(r["timesRun"] for r in res) / npairs,
            "pairs_with_nan": int(sum(r["status"] != 0 for r in res)),
            "lane_steps": steps, "lane_steps_per_s": steps / dt, "fp64_frac": insts / dt / FP64_LANE_OPS_PEAK,
            "fp64_frac_kernel_time": insts / (tm["like_ms"] / 1e3) / FP64_LANE_OPS_PEAK,
            "launches": {"sweep": st["sweep_launches"], "chunked": st["chunked_launches"],
                         "sweep_ms": st["sweep_ms"], "chunked_ms": st["chunked_ms"],
                         "sweep_lane_steps": st["sweep_point_steps"], "chunked_lane_steps": st["chunked_point_steps"]},
            "note": note or "relate_hip_fit, options.all.pair.txt semantics (calcA, timesToRun 8: the 2-parameter mode whose "
                    "runs the reference itself keeps on their trajectory in fewer than one case in ten under last-bit noise in "
                    "its objective: validated per run against that measured envelope, tests/test_gpu_sensitivity.py); not "
                    "part of value"}


def opts_linkage():
    """data/options.linkage.txt after the float parse: 1-parameter mode (k2 fixed at 0, a from k1), back = 50 -- what
    the manual prescribes for 500k-SNP chips -- and LD pruning at r^2 > 0.1."""
    f = lambda x: float(np.float32(x))  # noqa: E731
    return dict(min=f(0.025), epsilon=f(0.01), prune_val=f(0.1), alim0=f(0.001), alim1=f(0.15), fixA_val=f(0.5),
                fixK_val=0.0, phi=f(0.013), convTol=f(0.1), par=(f(0.3), 0.0, 0.5), back=50, doPrune=1, LD=1, ld_adj=1,
                double_recom=0, fixA=0, fixK=1, calcA=1, timesToConverge=3, timesToRun=1, doPar=0, seed_mode=0,
                seed=20240601)


def linkage_leg(local, inds, snps, allp, npairs):
    """Outside the timed region and not part of `value`: config 5's semantics (options.linkage.txt: back = 50, the generic
    staging route; pruning; one 1-parameter search per pair; decode of the one row the .post file keeps) at config 4's
    SNP count -- staging, fit and decode of one pair block, each timed on its own."""
    import relate_amd
    from relate_amd import synth
    opt = opts_linkage()
    geno, pos, chr_, _ = synth.make_dataset(inds, snps, device="cuda:%d" % local)
    H = relate_amd.Relate(local)
    t0 = time.perf_counter()
    M = H.init_globals(geno, pos, chr_, opt)
    t_pre = time.perf_counter() - t0
    del geno
    blk = allp[:npairs]
    H.bench_like(blk, eval_points(npairs, 1), 1)             # stages the block (generic K4) + one likelihood point
    stage_ms = H.timings()["stage_ms"]
    t0 = time.perf_counter()
    res = H.fit(blk)
    t_fit = time.perf_counter() - t0
    par = np.array([(r["a"], r["k2"], r["k1"]) for r in res])
    t0 = time.perf_counter()
    H.decode_f32(blk[:min(npairs, 1024)], par[:min(npairs, 1024)], rows=1)
    t_dec = time.perf_counter() - t0
    tm = H.timings()
    out = {"pairs": int(npairs), "snps_after_maf_and_pruning": int(M), "load_precalc_prune_s": t_pre,
           "stage_ms": stage_ms, "stage_gbs": ((REC + 0.75) * npairs * M) / (stage_ms * 1e-3) / 1e9 if stage_ms else None,
           "fit_seconds": t_fit, "fit_pairs_per_s": npairs / t_fit, "fit_like_kernel_seconds": tm["like_ms"] / 1e3,
           "decode_rows1_pairs": int(min(npairs, 1024)), "decode_rows1_seconds_with_host_copy": t_dec,
           "pairs_not_ok": int(sum(r["status"] != 0 for r in res)),
           "note": "options.linkage.txt semantics on the same synthetic matrix; not part of value"}
    H.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--inds", type=int, default=2000)
    ap.add_argument("--snps", type=int, default=500000)
    ap.add_argument("--pairs", type=int, default=12288,
                    help="pairs per block per GPU: 12 288 x 5e5 SNPs = 98 GB of records + 98 GB of posterior rows + 32 GB of "
                         "tile checkpoints of the 288 GB; E x pairs = 61 440 lanes, enough for the likelihood to run as whole "
                         "sweeps (rounds 1 and 2 took 5 120, whose 25 600 lanes had to be cut along the SNP axis)")
    ap.add_argument("--points", type=int, default=5, help="evaluation points per pair per like launch")
    ap.add_argument("--no-cpu", action="store_true", help="bench line only: no cpu_baseline, no fit leg")
    ap.add_argument("--no-fit", action="store_true")
    ap.add_argument("--no-baseline", action="store_true", help="skip the CPU baselines only (profiling runs)")
    ap.add_argument("--no-linkage", action="store_true")
    ap.add_argument("--linkage-pairs", type=int, default=4096)
    ap.add_argument("--fit-pairs", type=int, default=8192)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # Rehearsal switches for a one-GPU box (the driver's multi-GPU runs use neither): RELATE_BENCH_ONE_GPU=1 puts
    # every rank on device 0, RELATE_BENCH_BACKEND=gloo moves the two tiny collectives to the CPU.
    if os.environ.get("RELATE_BENCH_ONE_GPU") == "1":
        local = 0
    backend = os.environ.get("RELATE_BENCH_BACKEND", "nccl")
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    import relate_amd
    from relate_amd import synth

    opt = opts_all_pair()
    opt["back2"] = 2 * opt["back"]
    t_gen = time.perf_counter()
    geno, pos, chr_, _ = synth.make_dataset(args.inds, args.snps, device="cuda:%d" % local)
    t_gen = time.perf_counter() - t_gen
    H = relate_amd.Relate(local)
    t_pre = time.perf_counter()
    M = H.init_globals(geno, pos, chr_, opt)
    t_pre = time.perf_counter() - t_pre
    del geno

    all_pairs = synth.all_pairs(args.inds)
    b, e = relate_amd.shard(len(all_pairs), rank, world)
    mine = all_pairs[b:e]
    P, E = args.pairs, args.points
    nblocks = (len(mine) // P)
    if nblocks < 2:
        raise SystemExit("shard too small for --pairs")
    par = eval_points(P, E)
    par1 = np.ascontiguousarray(par[:, 0, :])
    sums = {"stage": 0.0, "like": 0.0, "decode_fwd": 0.0, "decode_bwd": 0.0}
    lastlike = None

    def step(i, record):
        blk = mine[(i % nblocks) * P:(i % nblocks + 1) * P]
        ms_like = H.bench_like(blk, par, 1)      # stages the block (K4) then K5
        st_ms = H.timings()["stage_ms"]
        f, bw = H.bench_decode(blk, par1, 1, rows=ROWS)     # K6 on the staged block
        if record:
            sums["stage"] += st_ms
            sums["like"] += ms_like
            sums["decode_fwd"] += f
            sums["decode_bwd"] += bw

    # The path's one exchange, through the C ABI: relate_hip_gather of the per-pair result records (here: one block's
    # worth per rank) over RCCL on the handle's stream, its communicator made once per handle from a rendezvous file
    # in a directory rank 0 names.  (Rehearsal on a one-GPU box: RCCL refuses two ranks on one device, so the records
    # go through the file transport of the same entry point.)
    from relate_amd import api
    rdv = None
    if dist is not None:
        import tempfile
        box = [tempfile.mkdtemp(prefix="relate_bench_rdv_") if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, device=torch.device(coll_dev if coll_dev == "cpu" else "cuda:%d" % local))
        rdv = os.path.join(box[0], "rdv")
    transport = 1 if (world > 1 and os.environ.get("RELATE_BENCH_ONE_GPU") == "1") else 0
    recs = np.zeros(P, dtype=api.fit_dtype())
    recs["negloglik"] = 1000.0 + rank
    recs["nsnp"] = M

    def exchange():
        got = api.gather_records(H, recs, world * P, rank, world, rdv, transport)
        assert got[(world - 1) * P].negloglik == 1000.0 + world - 1 and got[0].nsnp == M
        return H.gather_ranks()

    for i in range(args.warmup):
        step(i, False)
    exchange()                  # makes the communicator (once per handle, as in a scan)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, True)
    ranks_seen = exchange()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        K = args.steps
        hmm_steps = float(world) * K * P * M * (E + 2)
        per = {k: v / K for k, v in sums.items()}  # average ms per launch, HIP events on the library's stream
        lc, dc = H.last_chunks()
        BYTES = algorithmic_bytes(lc, dc)
        units = P * M
        gbs = {k: BYTES[k] * units / (per[k] * 1e-3) / 1e9 for k in per}
        # the dominant kernel of the step among the HBM-bound ones (the likelihood group is FP64-VALU-bound: roofline_fp64)
        dom = max((k for k in per if k != "like"), key=per.get)
        nv_like = 2 if E % 2 == 0 else 1          # points of one pair per lane (nv_for_points in relate_hip.hip)
        two_sided = lc == 1 and P * E // nv_like <= (98304 if nv_like == 1 else 65536)   # launch_like's route
        kernel_name = {"stage": "stage_pairs_win_kernel",
                       "like": "chunk_matrix_kernel<false, false>" if lc > 1 else
                               ("like_halves_kernel + like_join_kernel (two-sided sweep)" if two_sided else "like_kernel"),
                       "decode_fwd": "chunk_matrix_kernel<false, true>",
                       "decode_bwd": "decode_bwd_kernel"}
        traffic = pmc_traffic(P, M, E)
        bwd_key = "relate::decode_bwd_kernel<false, double, %d>" % ROWS
        tkey = {"stage": ["relate::stage_pairs_win_kernel<false>"], "like": ["relate::chunk_matrix_kernel<false, false>"] if lc > 1 else ["relate::like_halves_kernel<false, 4, 1>"],
                "decode_fwd": ["relate::chunk_matrix_kernel<false, true>"], "decode_bwd": [bwd_key]}[dom]
        tr_dom = sum(traffic[k] for k in tkey) if traffic and all(k in traffic for k in tkey) else None
        k6_keys = ["relate::chunk_matrix_kernel<false, true>", "relate::decode_combine_kernel", bwd_key]  # the launches `achieved` is over
        tr = sum(traffic[k] for k in k6_keys) if traffic and all(k in traffic for k in k6_keys) else None
        like_steps_s = P * M * E / (per["like"] * 1e-3)
        valu = VALU_PER_STEP[3 if lc > 1 else 1]
        dec_ms = per["decode_fwd"] + per["decode_bwd"]
        survey_gbs = SURVEY_BYTES["decode"] * units / (dec_ms * 1e-3) / 1e9
        line = {
            "metric": "pair x SNP fwd-bwd HMM steps/sec (node), all-pairs", "value": hmm_steps / dt,
            "unit": "pair*SNP HMM steps/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "value_note": "the (E+2) blend of the step: E likelihood sweeps + decode forward + decode backward per (pair, SNP); "
                          "SURVEY 8(d)'s three rates are like_point_steps_per_s, decode_pair_snp_per_s and "
                          "end_to_end_pair_snp_per_s below.  Kernel-resident: posteriors stay in HBM; with the posterior rows "
                          "copied to the host the step is PCIe-bound (tools/pcie_rate.py, DESIGN.md section 6)",
            "like_point_steps_per_s": world * like_steps_s,
            "decode_pair_snp_per_s": world * units / (dec_ms * 1e-3),
            "end_to_end_pair_snp_per_s": None,
            "gather": {"entry_point": "relate_hip_gather", "transport": "rccl" if transport == 0 else "file",
                       "n_ranks_seen": ranks_seen, "records": world * P},
            "config": {"workload": "config 4 synthetic all-pairs, pair block per step: %d individuals x %d SNPs "
                                   "(%d after MAF filter), %d pairs/GPU/step, E=%d like points + decode, "
                                   "options.all.pair.txt semantics (calcA, back=5)" % (args.inds, args.snps, M, P, E),
                       "parallelism": "pair-block shards x%d" % world,
                       "snp_chunks": {"like": lc, "decode": dc}},
            "roofline": {"bound": "hbm", "kernel": "K6 decode = chunk_matrix_kernel<false, true> + decode_combine_kernel + "
                                                   "decode_bwd_kernel (dominant launch: %s)" % kernel_name[dom],
                         "achieved": survey_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": survey_gbs / HBM_PEAK_GBS,
                         "traffic": tr, "traffic_dominant_launch": tr_dom,
                         "algorithmic_bytes": SURVEY_BYTES["decode"] * units,
                         "algorithmic_bytes_per_pair_snp": SURVEY_BYTES["decode"],
                         "decode_ms": dec_ms,
                         "frac_design": gbs[dom] / HBM_PEAK_GBS, "achieved_design": gbs[dom],
                         "algorithmic_bytes_design": BYTES[dom] * units,
                         "bytes_per_pair_snp_step": sum(BYTES.values()),
                         "note": "achieved / frac price the whole K6 (both passes) at SURVEY 8(d)'s algorithmic bytes, "
                                 "0.5 + 8 x rows per pair and SNP with rows = %d, over the HIP-event time of its launches; "
                                 "frac_design = the dominant launch of the step among the HBM-bound ones on this design's own "
                                 "bytes (materialised 16 B records, tile checkpoints); traffic = HBM bytes of one K6 (its three "
                                 "launches; traffic_dominant_launch: of the dominant one) from the committed PMC passes "
                                 "(profiles/pmc_traffic.json), null if they were taken on another configuration.  The likelihood kernels are FP64-VALU-bound: roofline_fp64" % ROWS},
            "roofline_fp64": {"kernel": kernel_name["like"], "valu_insts_per_lane_step": valu,
                              "achieved": like_steps_s * valu / 1e12, "peak": FP64_LANE_OPS_PEAK / 1e12,
                              "unit": "T lane-instructions/s", "frac": like_steps_s * valu / FP64_LANE_OPS_PEAK,
                              "frac_single_vector": like_steps_s * VALU_PER_STEP[1] / FP64_LANE_OPS_PEAK,
                              "lane_steps_per_s": like_steps_s,
                              "note": "frac counts every vector instruction of the route this launch takes (the chunked route "
                                      "carries three row vectors per lane); frac_single_vector prices the same point-steps at the "
                                      "27.3 instructions a single-vector sweep needs (about SURVEY 8(d)'s 15 FMA + 2 exp)"},
            "kernels_ms": per,
            "kernels_gbs": gbs,
            "kernels_hbm_frac": {k: v / HBM_PEAK_GBS for k, v in gbs.items()},
            "setup_s": {"synthetic_data": t_gen, "load_precalc": t_pre},
        }
        if world == 1 and not args.no_cpu:
            # The legs below are not part of `value`.  The step's buffers (posterior rows, checkpoints: 130 GB) are of no use
            # to them and the fit wants the memory for records: the handle is closed and the legs run on a fresh one.
            H.close()
            torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu and not args.no_linkage:
            line["linkage_mode"] = linkage_leg(local, args.inds, args.snps, all_pairs, min(args.linkage_pairs, len(all_pairs)))
        if world == 1 and not args.no_cpu and (not args.no_fit or not args.no_baseline):
            geno, pos, chr_, _ = synth.make_dataset(args.inds, args.snps, device="cuda:%d" % local)
            H = relate_amd.Relate(local)
            assert H.init_globals(geno, pos, chr_, opt) == M
            del geno
        if world == 1 and not args.no_cpu and not args.no_fit:
            line["fit"] = fit_leg(H, mine, min(args.fit_pairs, len(mine)), M)
            line["end_to_end_pair_snp_per_s"] = line["fit"]["pair_snp_per_s"]
            if args.fit_pairs > 0 and len(mine) >= 3 * args.fit_pairs + 4 * args.fit_pairs:
                # what a scan's later blocks run at: a block of the size the records allow (twice the first one here), on
                # the same handle, which has seen that first waves of restarts do not settle pairs in this mode
                line["fit_next_block"] = fit_leg(H, mine, 2 * args.fit_pairs, M, first=3 * args.fit_pairs,
                                                 note="the next block of the same scan, twice the pairs: all restarts of a "
                                                      "pair start together from the first round (the handle's previous fit saw "
                                                      "that first waves do not settle pairs); not part of value")
                line["end_to_end_pair_snp_per_s"] = line["fit_next_block"]["pair_snp_per_s"]
        if world == 1 and not args.no_cpu and not args.no_baseline:
            line["cpu_baseline"] = cpu_baseline(H, opt, E)
            line["cpu_baseline_all_cores"] = cpu_baseline_threads(H, opt, E)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
