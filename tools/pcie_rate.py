#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path: the same stage + like + decode step as bench.py, but through the host-buffer
entry points (relate_hip_like / relate_hip_decode), i.e. with the parameter upload and the posterior copy-back."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, relate_amd
from relate_amd import synth
P, E, N, M = 512, 5, 2000, 500000
opt = bench.opts_all_pair(); opt["back2"] = 2 * opt["back"]
geno, pos, chr_, _ = synth.make_dataset(N, M, device="cuda:0")
H = relate_amd.Relate(0)
M = H.init_globals(geno, pos, chr_, opt)
allp = synth.all_pairs(N)
par = bench.eval_points(P, E)
pin = H.post_buffer(P)
for it in range(6):
    blk = allp[(it % 3) * P:(it % 3 + 1) * P]
    pinned = it >= 3          # first three steps: ordinary numpy memory; last three: page-locked buffer
    t0 = time.perf_counter()
    H.like(blk, par)
    t1 = time.perf_counter()
    post, _, _ = H.decode(blk, np.ascontiguousarray(par[:, 0, :]), out=pin if pinned else None)
    dt = time.perf_counter() - t0
    print("P=%d M=%d %s: %.1f ms per step with host buffers (decode + copy-back %.1f ms) = %.3g HMM steps/s "
          "(posterior copy-back %.2f GB)" % (P, M, "pinned" if pinned else "pageable", dt * 1e3,
                                             (time.perf_counter() - t1) * 1e3, P * M * (E + 2) / dt,
                                             post.nbytes / 1e9), flush=True)
pin2 = H.post_buffer(P, 2)
for it in range(6):
    blk = allp[(it % 3) * P:(it % 3 + 1) * P]
    pinned = it >= 3
    t0 = time.perf_counter()
    H.like(blk, par)
    post, _ = H.decode_rows(blk, np.ascontiguousarray(par[:, 0, :]), 2, out=pin2 if pinned else None)
    dt = time.perf_counter() - t0
    print("P=%d M=%d double rows IBD0+IBD1 (relate_hip_decode_rows), %s: %.1f ms per step = %.3g HMM steps/s (copy-back "
          "%.2f GB)" % (P, M, "pinned" if pinned else "pageable", dt * 1e3, P * M * (E + 2) / dt, post.nbytes / 1e9), flush=True)
pin32 = H.post_buffer(P, 2, np.float32)
for it in range(3):
    blk = allp[it * P:(it + 1) * P]
    t0 = time.perf_counter()
    H.like(blk, par)
    post, _ = H.decode_f32(blk, np.ascontiguousarray(par[:, 0, :]), 2, out=pin32)
    dt = time.perf_counter() - t0
    print("P=%d M=%d float32 rows IBD0+IBD1, pinned: %.1f ms per step = %.3g HMM steps/s (copy-back %.2f GB)" % (
        P, M, dt * 1e3, P * M * (E + 2) / dt, post.nbytes / 1e9), flush=True)
