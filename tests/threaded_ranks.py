"""P ranks of the product as P threads of ONE process on ONE GPU (test infrastructure).

The container and the GPU box have one GPU, and RCCL refuses two ranks on one device,
so this is how the multi-rank data path -- gather of subtree roots, all-gather,
scatter, replicated top sweep, CAAS all-reduce -- runs on real hardware with P > 3:
every rank is a thread with its own cedr::mpi::Parallel (callbacks transport); the
collectives meet at a barrier and exchange host copies. ctypes releases the GIL
around library calls, so ranks really interleave on the device.
"""
import threading

import numpy as np


class ThreadComm:
    def __init__(self, nranks, timeout=600):
        self.P = nranks
        self.bar = threading.Barrier(nranks, timeout=timeout)
        self.slots = [None] * nranks
        self.cv = threading.Condition()
        self.fifo = {}
        self.timeout = timeout

    def allgather(self, rank, x):
        self.slots[rank] = np.array(x, dtype=np.float64)
        self.bar.wait()
        out = np.concatenate(self.slots)
        self.bar.wait()
        return out

    def allreduce(self, rank, x):
        self.slots[rank] = np.array(x, dtype=np.float64)
        self.bar.wait()
        out = self.slots[0].copy()
        for q in range(1, self.P):  # fixed order: identical bits on every rank
            out += self.slots[q]
        self.bar.wait()
        return out

    def exchange(self, rank, sends, recvs):
        """One round of point-to-point messages (level mode): FIFO per (source, destination) --
        the rounds come in the same order on both sides, with at most one message per pair."""
        for peer, a in sends:
            with self.cv:
                self.fifo.setdefault((rank, peer), []).append(np.array(a, dtype=np.float64))
                self.cv.notify_all()
        out = []
        for peer, n in recvs:
            with self.cv:
                ok = self.cv.wait_for(lambda: self.fifo.get((peer, rank)) or self.bar.broken, self.timeout)
                if not ok or self.bar.broken:
                    raise threading.BrokenBarrierError()
                a = self.fifo[(peer, rank)].pop(0)
            assert a.size == n, (a.size, n)
            out.append(a)
        return out

    def parallel(self, ca, rank):
        return ca.Parallel.host_staged_p2p(rank, self.P, lambda x: self.allgather(rank, x),
                                           lambda x: self.allreduce(rank, x),
                                           lambda sends, recvs: self.exchange(rank, sends, recvs))


class RcclThreads(ThreadComm):
    """The same ranks-as-threads harness over the product's OWN RCCL communicator class: every thread
    makes ca.Parallel.rccl(id, rank, P). Only meaningful when the process binds the tests' double of RCCL
    (CEDR_HIP_RCCL_LIB = tests/cpp/libfake_rccl.so): the real one refuses two ranks on one device."""

    def __init__(self, ca, nranks, timeout=600):
        ThreadComm.__init__(self, nranks, timeout)
        self.uid = ca.Parallel.rccl_unique_id()

    def parallel(self, ca, rank):
        return ca.Parallel.rccl(self.uid, rank, self.P)


def run_ranks(nranks, body, timeout=600, comm=None):
    """body(rank, comm) on nranks threads; re-raises the first failure."""
    comm = comm or ThreadComm(nranks, timeout)
    errs = [None] * nranks
    res = [None] * nranks

    def wrap(r):
        try:
            res[r] = body(r, comm)
        except BaseException as e:  # noqa: BLE001 -- reported to the caller below
            errs[r] = e
            comm.bar.abort()
            with comm.cv:
                comm.cv.notify_all()
    ts = [threading.Thread(target=wrap, args=(r,)) for r in range(nranks)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    real = [e for e in errs if e is not None and not isinstance(e, threading.BrokenBarrierError)]
    if real:
        # (a rank that fails aborts the barrier, and the others then fail inside their next collective: show every rank's
        # error, the cause is the one that does not mention a collective)
        import sys
        for r, e in enumerate(errs):
            if e is not None:
                print("run_ranks: rank %d: %s: %s" % (r, type(e).__name__, str(e)[:400].replace("\n", " ")), file=sys.stderr)
        raise real[0]
    if any(errs):
        raise [e for e in errs if e][0]
    return res


def run_qlt(ca, nranks, ncells, types, v, prefer=False, pseudorandom=False, imbalanced=False, nrepeat=1,
            partial=False, want_level_mode=None, comm=None):
    """Global arrays in, global result out; every rank owns what Tree.make_1d gives it.
    partial: every rank passes only its part of the tree (tree::Node::level)."""
    out = np.zeros_like(v.Qm)

    def body(rank, comm):
        tree = ca.Tree.make_1d(ncells, nranks, pseudorandom, imbalanced)
        if partial:
            tree = tree.prune(rank)
        q = ca.QLT(comm.parallel(ca, rank), ncells, tree, prefer)
        for t in types:
            q.declare_tracer(t, 0)
        q.end_tracer_declarations()
        q.finish_setup()
        if want_level_mode is not None:
            assert q.plan()["level_mode"] == want_level_mode
        g = q.get_owned_glblcells()
        d = [ca.DeviceArray.from_numpy(a) for a in
             (v.rhom[g], v.Qm[:, g], v.Qm_min[:, g], v.Qm_max[:, g], v.Qm_prev[:, g])]
        q.set_rhom(d[0])
        for _ in range(nrepeat):
            q.set_Qm(d[1], d[2], d[3], d[4])
            q.run()
        od = ca.DeviceArray(len(types) * len(g))
        q.get_Qm(od)
        ca.synchronize()
        out[:, g] = od.numpy((len(types), len(g)))
        return len(g)
    counts = run_ranks(nranks, body, comm=comm)
    assert sum(counts) == ncells
    return out


def run_caas(ca, nranks, ncells, types, v, comm=None):
    """Contiguous split of the global arrays; returns the global result."""
    out = np.zeros_like(v.Qm)
    bounds = [ncells * r // nranks for r in range(nranks + 1)]

    def body(rank, comm):
        sl = slice(bounds[rank], bounds[rank + 1])
        n = bounds[rank + 1] - bounds[rank]
        c = ca.CAAS(comm.parallel(ca, rank), n)
        for t in types:
            c.declare_tracer(t, 0)
        c.end_tracer_declarations()
        c.finish_setup()
        d = [ca.DeviceArray.from_numpy(a) for a in
             (v.rhom[sl], v.Qm[:, sl], v.Qm_min[:, sl], v.Qm_max[:, sl], v.Qm_prev[:, sl])]
        c.set_rhom(d[0])
        c.set_Qm(d[1], d[2], d[3], d[4])
        c.run()
        od = ca.DeviceArray(len(types) * n)
        c.get_Qm(od)
        ca.synchronize()
        out[:, sl] = od.numpy((len(types), n))
    run_ranks(nranks, body, comm=comm)
    return out, [bounds[r + 1] - bounds[r] for r in range(nranks)]
