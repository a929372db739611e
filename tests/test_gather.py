"""relate_hip_gather (include/relate_hip.h), the path's one exchange, without a GPU: the file transport moves the
per-pair result records of N processes through a directory every rank sees; every rank must end with all records in
pair order, for shard sizes that differ by one and for more ranks than records.  (The RCCL transport shares the
bookkeeping -- relate_hip_shard offsets, shards padded to the longest -- and runs in tests/test_cli_gpu.py.)"""
import ctypes as C
import multiprocessing as mp
import os

import numpy as np
import pytest

import relate_amd
from relate_amd import api


def _records(P):
    out = (api.RelateFit * max(P, 1))()
    for p in range(P):
        out[p].a, out[p].k2, out[p].k1, out[p].k0 = 0.01 * p, 0.001 * p, 0.5, 0.25
        out[p].negloglik, out[p].uLike, out[p].poLike = 1000.0 + p, 2000.0 + p, 3000.0 + p
        out[p].timesRun, out[p].timesConverged, out[p].nsnp, out[p].status = p % 8, p % 3, 400 + p, 0
    return out


def _rank(rank, n, P, rdv, q):
    try:
        L = relate_amd.load_library()
        L.relate_hip_gather.argtypes = [C.c_void_p, C.POINTER(api.RelateFit), C.c_int64, C.c_int, C.c_int, C.c_char_p,
                                        C.c_int, C.POINTER(api.RelateFit)]
        b, e = relate_amd.shard(P, rank, n)
        full = _records(P)
        mine = (api.RelateFit * max(e - b, 1))(*[full[p] for p in range(b, e)])
        got = (api.RelateFit * max(P, 1))()
        rc = L.relate_hip_gather(None, mine, P, rank, n, rdv.encode(), 1, got)
        q.put((rank, rc, bytes(got)[:C.sizeof(api.RelateFit) * P] == bytes(full)[:C.sizeof(api.RelateFit) * P]))
    except Exception as ex:  # noqa: BLE001
        q.put((rank, -99, repr(ex)))


@pytest.mark.parametrize("n,P", [(2, 7), (3, 190), (4, 2), (1, 5)])
def test_file_transport_gathers_every_shard_in_pair_order(tmp_path, n, P):
    os.environ["RELATE_GATHER_TIMEOUT"] = "60"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rank, args=(r, n, P, str(tmp_path / "rdv"), q)) for r in range(n)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == [(r, 0, True) for r in range(n)], res


def test_gather_rejects_bad_arguments():
    L = relate_amd.load_library()
    L.relate_hip_gather.argtypes = [C.c_void_p, C.POINTER(api.RelateFit), C.c_int64, C.c_int, C.c_int, C.c_char_p,
                                    C.c_int, C.POINTER(api.RelateFit)]
    r = _records(4)
    assert L.relate_hip_gather(None, r, 4, 2, 2, b"x", 1, r) < 0       # rank out of range
    assert L.relate_hip_gather(None, r, 4, 0, 2, None, 1, r) < 0       # several ranks need a rendezvous path
    assert L.relate_hip_gather(None, r, 4, 0, 1, None, 0, r) < 0       # the RCCL transport needs a handle
    assert b"handle" in L.relate_hip_last_error(None)
