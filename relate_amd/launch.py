"""Launcher of the C++ host driver on the GPUs of one node: one relateHMM_hip process per GPU, rank r on GPU r.

    python -m relate_amd.launch -n 8 -- -o options.txt -plink data -d keep -post out.post -k out.k -est out.est

Every rank runs `relateHMM_hip <args> -rank r -nranks n`: it takes the contiguous shard of the pair list that
relate_hip_shard gives it, writes its posterior rows to out.post.rank<r>, and the estimates are gathered over RCCL
(relate_hip_gather; `-gather file` on the command line moves them through files instead); rank 0 writes out.k / out.est
for all pairs.  No numerics here: this module only starts processes and waits for them.  The first rank that ends
badly (non-zero exit, or killed by a signal: a GPU fault aborts with SIGABRT) ends the run: the other ranks would wait
for it in the gather for ever, so they are terminated, and the launcher's exit status is non-zero."""
import argparse
import os
import subprocess
import sys
import time

CLI = os.path.join(os.path.dirname(os.path.abspath(__file__)), "relateHMM_hip")


def main(argv=None):
    ap = argparse.ArgumentParser(prog="relate_amd.launch")
    ap.add_argument("-n", "--ranks", type=int, required=True, help="processes = GPUs of this node to use")
    ap.add_argument("rest", nargs=argparse.REMAINDER, help="-- followed by relateHMM_hip's arguments")
    a = ap.parse_args(argv)
    rest = a.rest[1:] if a.rest[:1] == ["--"] else a.rest
    if not os.path.exists(CLI):
        raise SystemExit("relateHMM_hip is not built: python -c 'import __graft_entry__ as g; g.build()'")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([CLI] + rest + ["-rank", str(r), "-nranks", str(a.ranks)],
                              env=dict(env, RANK=str(r), WORLD_SIZE=str(a.ranks), LOCAL_RANK=str(r)))
             for r in range(a.ranks)]
    return wait_all(procs)


def exit_status(code):
    """A child's returncode as a shell would report it: 128 + signal number for a rank killed by a signal."""
    return code if code >= 0 else 128 - code


def wait_all(procs, poll=0.1, grace=10.0):
    """Waits for every rank; as soon as one ends with a non-zero status the remaining ones are terminated (SIGTERM,
    SIGKILL after `grace` seconds).  Returns 0 only if every rank returned 0, else the status of the first failure."""
    failed = 0
    live = list(procs)
    while live and not failed:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and not failed:
                failed = exit_status(code)
        if live and not failed:
            time.sleep(poll)
    if failed:
        for p in live:
            p.terminate()
        t0 = time.time()
        for p in live:
            try:
                p.wait(timeout=max(0.0, grace - (time.time() - t0)))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return failed


if __name__ == "__main__":
    sys.exit(main())
