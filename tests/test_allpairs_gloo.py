"""relate_amd/allpairs.py with world_size 2 on CPU (gloo): real option file, real input readers, real sharding,
real gather and writers; the GPU handle is replaced by a stand-in whose "estimates" are a function of the pair, so
the merged outputs can be checked exactly.  (The kernels are covered by the -m gpu tests.)"""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "data")


class FakeHandle:
    def init_globals(self, geno, pos, chr_, opt):
        self.M = 7
        return self.M

    def sgl(self, pairs, want_post=True, **kw):
        res, post = [], np.zeros((len(pairs), 3, self.M))
        for n, (i, j) in enumerate(pairs):
            res.append(dict(a=0.01 * i, k2=0.0, k1=(i * 100 + j) / 1e4, k0=0.0, negloglik=float(i + j), uLike=0.0,
                            poLike=0.0, timesRun=1, timesConverged=1, nsnp=self.M, status=0))
            post[n] = (i * 100 + j) / 1e4
            post[n][:, 3] = -1
        return res, (post if want_post else None), None

    def decode(self, pairs, params, **kw):
        post = np.zeros((len(pairs), 3, self.M))
        for n, (i, j) in enumerate(pairs):
            assert abs(params[n][2] - (i * 100 + j) / 1e4) < 1e-15   # decoded at the fitted k1 of that pair
            post[n] = (i * 100 + j) / 1e4
            post[n][:, 3] = -1
        return post, None, np.full(len(pairs), self.M, dtype=np.int32)

    def decode_f32(self, pairs, params, rows=2, out=None):
        post, _, ns = self.decode(pairs, params)
        return np.ascontiguousarray(post[:, [2, 1, 0][:rows], :], dtype=np.float32), ns


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, extra=()):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from relate_amd import allpairs
    from test_allpairs_gloo import FakeHandle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a = allpairs.parse_args(["-o", DATA + "/options.linkage.txt", "-file", DATA + "/dat_small", "-d",
                             DATA + "/dat_small.keep", "-out", out, "-seed", "1", "-block", "17"] + list(extra))
    allpairs.run(a, FakeHandle, dist, torch.device("cpu"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allpairs_outputs(tmp_path):
    out = os.path.join(tmp_path, "run")
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from relate_amd import io as rio
    pairs = rio.keep_list_pairs(DATA + "/dat_small.keep")
    assert len(pairs) == 190
    k = [float(v) for v in open(out + ".k").read().split()]
    want = [float("%.4g" % ((i * 100 + j) / 1e4)) for i, j in pairs]
    assert k == want                                           # fixK2 = 0: one k1 line per pair, in pair order
    est = [line.split("\t") for line in open(out + ".est").read().splitlines()]
    assert [(int(e[0]), int(e[1])) for e in est] == [tuple(p) for p in pairs]
    rows = open(out + ".post.rank0").read().splitlines() + open(out + ".post.rank1").read().splitlines()
    assert len(rows) == 190                                    # one IBD0 row per pair, shards concatenate in order
    for (i, j), row in zip(pairs, rows):
        t = row.split("\t")
        assert t[-1] == "" and len(t) == 8 and t[3] == "-1" and float(t[0]) == float("%.4g" % ((i * 100 + j) / 1e4))


def test_two_rank_binary_shards_with_post_subblocks(tmp_path):
    """fit blocks of 17 pairs, posteriors decoded 5 pairs at a time, written as float32 shards"""
    import json
    out = os.path.join(tmp_path, "runb")
    port = _free_port()
    ctx = mp.get_context("spawn")
    extra = ("-post-block", "5", "-post-format", "bin")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out, extra)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from relate_amd import io as rio
    pairs = rio.keep_list_pairs(DATA + "/dat_small.keep")
    rows = []
    first = []
    for r in range(2):
        meta = json.load(open("%s.postbin.rank%d.json" % (out, r)))
        assert meta["snps"] == 7 and meta["rows_per_pair"] == 1
        first.append(meta["first_pair"])
        x = np.fromfile("%s.postbin.rank%d" % (out, r), dtype="<f4").reshape(meta["rows"], meta["snps"])
        rows.append(x)
    assert first == [0, 95]
    x = np.concatenate(rows)
    assert x.shape == (190, 7)
    want = np.array([(i * 100 + j) / 1e4 for i, j in pairs], dtype=np.float32)
    assert np.array_equal(x[:, 0], want) and np.all(x[:, 3] == -1)
