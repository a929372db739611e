"""Multi-GPU plumbing: pair-block sharding and the final gather of per-pair estimates.

The all-pairs path shards naturally: every rank owns a contiguous block of the ordered pair list
(relate_hip_shard) and replicated read-only inputs; there is no exchange while pairs are processed.  The only
collective is one all_gather of the per-pair result records at the end (RCCL over xGMI with backend "nccl";
"gloo" in the CPU tests).  One process per GPU, launched by torch.distributed.run.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import api

FIT_FIELDS = ("a", "k2", "k1", "k0", "negloglik", "uLike", "poLike", "timesRun", "timesConverged", "nsnp", "status")


def shard_pairs(pairs, rank=None, world=None):
    """Rows [begin, end) of `pairs` owned by this rank (contiguous, sizes differ by at most one)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    b, e = api.shard(len(pairs), rank, world)
    return b, e


def fits_to_array(fits):
    return np.array([[float(f[k]) for k in FIT_FIELDS] for f in fits], dtype=np.float64).reshape(-1, len(FIT_FIELDS))


def gather_rows(local, total_rows, device=None):
    """all_gather of row blocks of unequal length: returns the (total_rows x F) array in pair order on every rank."""
    world = dist.get_world_size()
    F = local.shape[1]
    sizes = [api.shard(total_rows, r, world) for r in range(world)]
    nmax = max(e - b for b, e in sizes)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    buf = torch.zeros((nmax, F), dtype=torch.float64, device=device)
    if len(local):
        buf[:len(local)] = torch.from_numpy(np.ascontiguousarray(local)).to(device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    full = np.zeros((total_rows, F))
    for r, (b, e) in enumerate(sizes):
        full[b:e] = out[r][:e - b].cpu().numpy()
    return full


def run_sharded(handle_factory, pairs, work):
    """Each rank applies `work(handle, my_pairs)` -> (n_local x F array) to its shard, then gathers all rows."""
    b, e = shard_pairs(pairs)
    H = handle_factory()
    local = work(H, pairs[b:e])
    return gather_rows(np.asarray(local, dtype=np.float64).reshape(e - b, -1), len(pairs))
