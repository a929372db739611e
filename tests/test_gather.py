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


def _records(P, shift=0.0):
    out = (api.RelateFit * max(P, 1))()
    for p in range(P):
        out[p].a, out[p].k2, out[p].k1, out[p].k0 = 0.01 * p + shift, 0.001 * p, 0.5, 0.25
        out[p].negloglik, out[p].uLike, out[p].poLike = 1000.0 + p, 2000.0 + p, 3000.0 + p
        out[p].timesRun, out[p].timesConverged, out[p].nsnp, out[p].status = p % 8, p % 3, 400 + p, 0
    return out


def _rank(rank, n, P, rdv, q, shift=0.0, delay=0.0, rounds=1):
    try:
        import time
        time.sleep(delay)
        L = relate_amd.load_library()
        L.relate_hip_gather.argtypes = [C.c_void_p, C.POINTER(api.RelateFit), C.c_int64, C.c_int, C.c_int, C.c_char_p,
                                        C.c_int, C.POINTER(api.RelateFit)]
        b, e = relate_amd.shard(P, rank, n)
        for rnd in range(rounds - 1):      # earlier gathers of the same process under the same names (h = NULL)
            full = _records(P, shift + 100 * (rnd + 1))
            mine = (api.RelateFit * max(e - b, 1))(*[full[p] for p in range(b, e)])
            got = (api.RelateFit * max(P, 1))()
            rc = L.relate_hip_gather(None, mine, P, rank, n, rdv.encode(), 1, got)
            if rc or bytes(got)[:C.sizeof(api.RelateFit) * P] != bytes(full)[:C.sizeof(api.RelateFit) * P]:
                q.put((rank, rc, "round %d" % rnd))
                return
        full = _records(P, shift)
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


def _run(tmp_path, n, P, **kw):
    os.environ["RELATE_GATHER_TIMEOUT"] = "60"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    delays = kw.pop("delays", [0.0] * n)
    ps = [ctx.Process(target=_rank, args=(r, n, P, str(tmp_path / "rdv"), q), kwargs=dict(kw, delay=delays[r]))
          for r in range(n)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    return res


def _leftovers(tmp_path):
    return sorted(f for f in os.listdir(tmp_path) if f.startswith("rdv"))


def test_a_second_run_under_the_same_rendezvous_name_does_not_see_the_first_runs_files(tmp_path):
    """Two runs in one directory with different records: the second must gather ITS records (round 2's advisor
    finding: a shard file of the right size left by the first run was taken for the second run's).  The exchange
    removes its own files, so the directory is clean after each run."""
    assert _run(tmp_path, 3, 190, shift=0.0) == [(r, 0, True) for r in range(3)]
    assert _leftovers(tmp_path) == []
    assert _run(tmp_path, 3, 190, shift=5.0) == [(r, 0, True) for r in range(3)]
    assert _leftovers(tmp_path) == []


def test_files_a_dead_run_left_behind_are_not_taken_for_this_runs(tmp_path):
    """What a run that died mid-gather leaves: shard files of exactly the right size (old format and new format with
    another session's nonce), a session that went GO, hellos and acks of ranks that no longer exist.  The live ranks
    start staggered, so rank 0 first reads the dead run's hello of rank 1 and rank 2 arrives last."""
    n, P = 3, 190
    rec = C.sizeof(api.RelateFit)
    magic = 0x333045544C4552
    for r in range(n):
        b, e = relate_amd.shard(P, r, n)
        stale = _records(P, 77.0)
        body = bytes((api.RelateFit * (e - b))(*[stale[p] for p in range(b, e)]))
        assert len(body) == rec * (e - b)
        hdr = np.array([magic, 0xDEAD, e - b], dtype=np.uint64).tobytes()
        (tmp_path / ("rdv.g0.fit.%d" % r)).write_bytes(hdr + body)
        (tmp_path / ("rdv.g0.done.%d" % r)).write_bytes(np.array([magic, 0xDEAD, 0], dtype=np.uint64).tobytes())
        if r == 0:
            continue                     # rank 0 writes neither
        (tmp_path / ("rdv.g0.hello.%d" % r)).write_bytes(np.array([magic, 0xBEEF + r], dtype=np.uint64).tobytes())
        (tmp_path / ("rdv.g0.ack.%d" % r)).write_bytes(np.array([magic, 0xDEAD, 0xBEEF + r], dtype=np.uint64).tobytes())
    sess = np.array([magic, 0xDEAD], dtype=np.uint64).tobytes() + np.array([1, n], dtype=np.int32).tobytes() + \
        np.array([P], dtype=np.int64).tobytes() + bytes(128) + np.array([0, 0xBEEF + 1, 0xBEEF + 2], dtype=np.uint64).tobytes()
    (tmp_path / "rdv.g0.session").write_bytes(sess)
    assert _run(tmp_path, n, P, shift=1.0, delays=[0.0, 0.4, 0.8]) == [(r, 0, True) for r in range(n)]
    assert _leftovers(tmp_path) == []


def test_consecutive_gathers_of_one_process_without_a_handle(tmp_path):
    """h = NULL: every gather of a process uses the same file names (no per-handle sequence number); three in a row
    with different records must each deliver their own."""
    assert _run(tmp_path, 2, 33, rounds=3) == [(r, 0, True) for r in range(2)]
    assert _leftovers(tmp_path) == []


def test_ranks_that_disagree_on_the_pair_count_are_told_so(tmp_path):
    os.environ["RELATE_GATHER_TIMEOUT"] = "5"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rank, args=(r, 2, 10 + r, str(tmp_path / "rdv"), q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(timeout=30)
    assert res[1][1] < 0, res          # rank 1 reads rank 0's session and sees another pair count
    os.environ["RELATE_GATHER_TIMEOUT"] = "60"


def test_launcher_reports_a_rank_killed_by_a_signal_and_ends_the_others():
    """relate_amd.launch.wait_all (round 2's advisor finding: max([-6, 0, 0]) is 0): a rank that dies of a signal makes
    the status non-zero, and ranks that would wait in the gather for ever are terminated."""
    import subprocess
    import sys
    import time
    from relate_amd import launch
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", "import os, signal; os.kill(os.getpid(), signal.SIGABRT)"]),
             subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"]),
             subprocess.Popen([sys.executable, "-c", "pass"])]
    assert launch.wait_all(procs) == 128 + 6
    assert time.time() - t0 < 60 and all(p.poll() is not None for p in procs)
    assert launch.wait_all([subprocess.Popen([sys.executable, "-c", "pass"]) for _ in range(2)]) == 0
    assert launch.wait_all([subprocess.Popen([sys.executable, "-c", "raise SystemExit(3)"])]) == 3


def test_gather_rejects_bad_arguments():
    L = relate_amd.load_library()
    L.relate_hip_gather.argtypes = [C.c_void_p, C.POINTER(api.RelateFit), C.c_int64, C.c_int, C.c_int, C.c_char_p,
                                    C.c_int, C.POINTER(api.RelateFit)]
    r = _records(4)
    assert L.relate_hip_gather(None, r, 4, 2, 2, b"x", 1, r) < 0       # rank out of range
    assert L.relate_hip_gather(None, r, 4, 0, 2, None, 1, r) < 0       # several ranks need a rendezvous path
    assert L.relate_hip_gather(None, r, 4, 0, 1, None, 0, r) < 0       # the RCCL transport needs a handle
    assert b"handle" in L.relate_hip_last_error(None)
