"""N > 1 path on CPU: world_size-2 gloo run of the pair-block sharding and the final gather (relate_amd/dist.py).
The per-shard work is a stand-in function of the pair indices (the GPU kernels are covered by the -m gpu tests);
what is checked is that the shards are contiguous, complete and gathered back in pair order."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, npairs, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from relate_amd import dist as rd, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pairs = synth.all_pairs(npairs)

    def work(_h, mine):
        return np.stack([mine[:, 0] * 1000.0 + mine[:, 1], np.full(len(mine), float(rank))], axis=1)
    full = rd.run_sharded(lambda: None, pairs, work)
    b, e = rd.shard_pairs(pairs)
    q.put((rank, b, e, full))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    world, n = 2, 9   # 36 pairs
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from relate_amd import synth
    pairs = synth.all_pairs(n)
    want = pairs[:, 0] * 1000.0 + pairs[:, 1]
    got.sort(key=lambda t: t[0])
    assert got[0][1] == 0 and got[0][2] == got[1][1] and got[1][2] == len(pairs)
    for rank, b, e, full in got:
        assert np.array_equal(full[:, 0], want)
        assert np.array_equal(full[:18, 1], np.zeros(18)) and np.array_equal(full[18:, 1], np.ones(18))
