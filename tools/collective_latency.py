#!/usr/bin/env python3
"""Latency of ONE collective of a cedr::mpi::Parallel transport, the way a step uses it: KB-sized, enqueued on a stream
between kernels. Ranks as processes on GPU 0 (the only way several ranks run on a one-GPU box; RCCL refuses two ranks on
a device, so at P > 1 only the peer-flag transport can be measured here).

  collective_latency.py                  one rank: self, rccl (size 1), peer-flags (size 1)
  collective_latency.py --ranks 2|3|4    starts that many processes over the peer-flag transport

Every rank enqueues N all-reduces (or all-gathers) of COUNT doubles back to back on one stream and divides the wall time
between two synchronisations by N: the critical path of one collective (put launch, the peers' stores and counter adds
becoming visible, the stream wait, the take launch), all ranks in lock step. Results are checked (sum / blocks of the
ranks' payloads). Prints one line per case."""
import argparse
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def measure(ca, par, rank, size, count, n, label):
    import numpy as np
    send = ca.DeviceArray.from_numpy(np.arange(count, dtype=np.float64) + 1000.0 * rank)
    rs = ca.DeviceArray(count)
    rg = ca.DeviceArray(count * size)
    out = []
    for kind in ("allreduce", "allgather"):
        fn = (lambda: par.allreduce_sum(send, rs, count)) if kind == "allreduce" else (lambda: par.allgather(send, rg, count))
        for _ in range(50):
            fn()
        ca.synchronize()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            ca.synchronize()
            dt = (time.perf_counter() - t0) / n * 1e6
            best = dt if best is None else min(best, dt)
        if kind == "allreduce":
            want = np.arange(count, dtype=np.float64) * size + 1000.0 * sum(range(size))
            ok = np.array_equal(rs.numpy(), want)
        else:
            want = np.concatenate([np.arange(count, dtype=np.float64) + 1000.0 * r for r in range(size)])
            ok = np.array_equal(rg.numpy(), want)
        out.append("%s %.2f us%s" % (kind, best, "" if ok else " WRONG RESULT"))
    print("%-34s rank %d of %d, %d doubles, %d calls back to back: %s" % (label, rank, size, count, n, "; ".join(out)),
          flush=True)
    return all("WRONG" not in o for o in out)


def file_gather(d, name, rank, size, blob, timeout=120):
    """Every rank's blob, in rank order, through files in d."""
    tmp = os.path.join(d, ".%s%d.tmp" % (name, rank))
    with open(tmp, "wb") as f:
        f.write(blob)
    os.rename(tmp, os.path.join(d, "%s%d" % (name, rank)))
    out, t0 = [], time.monotonic()
    for r in range(size):
        p = os.path.join(d, "%s%d" % (name, r))
        while not os.path.exists(p):
            if time.monotonic() - t0 > timeout:
                raise SystemExit("rank %d: no %s from rank %d" % (rank, name, r))
            time.sleep(0.01)
        with open(p, "rb") as f:
            out.append(f.read())
    return out


def rank_main(a):
    import compose_amd as ca
    mb, handle = ca.Parallel.peer_mailbox(a.ranks)
    handles = file_gather(a.dir, "handle", a.rank, a.ranks, handle)
    par = ca.Parallel.peer_flags(mb, handles, a.rank, a.ranks)
    file_gather(a.dir, "mapped", a.rank, a.ranks, b"1")
    ok = measure(ca, par, a.rank, a.ranks, a.count, a.calls, "peer-flags (processes on GPU 0)")
    file_gather(a.dir, "done", a.rank, a.ranks, b"1")
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=1)
    ap.add_argument("--count", type=int, default=160)
    ap.add_argument("--calls", type=int, default=2000)
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--dir", default="")
    a = ap.parse_args()
    if a.rank >= 0:
        return rank_main(a)
    if a.ranks > 1:
        if a.ranks > 4:
            raise SystemExit("at most 4 rank processes on the one GPU")
        with tempfile.TemporaryDirectory() as d:
            ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--ranks", str(a.ranks), "--rank", str(r),
                                    "--dir", d, "--count", str(a.count), "--calls", str(a.calls)]) for r in range(a.ranks)]
            rc = 0
            try:
                for p in ps:
                    rc = p.wait(timeout=300) or rc
            finally:
                for p in ps:  # exactly the children started above
                    if p.poll() is None:
                        p.kill()
            return rc
    import compose_amd as ca
    ok = measure(ca, ca.Parallel.self_(), 0, 1, a.count, a.calls, "single-rank (a copy)")
    ok &= measure(ca, ca.Parallel.rccl(ca.Parallel.rccl_unique_id(), 0, 1), 0, 1, a.count, a.calls, "rccl, communicator of ONE rank")
    mb, h = ca.Parallel.peer_mailbox(1)
    ok &= measure(ca, ca.Parallel.peer_flags(mb, [h], 0, 1), 0, 1, a.count, a.calls, "peer-flags, communicator of ONE rank")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
