"""One RANK of a P-rank job over the product's peer-flag transport (cedr_mpi.hpp: PeerFlagComm), ranks as PROCESSES on
the one GPU of the box (every process has its own hardware queues, so a rank's stream wait cannot hold up the launch of
another rank that releases it -- ranks as threads of one process would share queues). Started P times by
tests/test_gpu_peer_flags.py: usage  peer_flags_worker.py RANK SIZE DIR.

The mailbox handles go from rank to rank through files in DIR. Every rank then runs, over ONE communicator: QLT through
the all-gather (36 harness tracers, twice, two different sets of values in turn), CAAS's all-reduce, the BFB reducer
(inputs scaled by powers of two from call to call, so a stale slot or an early wake-up shows; one call at a time and
in bursts of 16 without host synchronisation), QLT through per-level point-to-point rounds (partial trees, forced, pseudorandom
ownership) and the reducer from a partial tree, QLT and CAAS alternating on two streams -- and compares ITS cells bit for bit with the single-rank oracle (CAAS's own
all-reduce: 1e-13 per tracer, and the same bits on every rank). A mismatch is recorded and reported at the end: a rank
never leaves early, since the others would then wait for its blocks. Test infrastructure."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]

import common  # noqa: E402
import compose_amd as ca  # noqa: E402
import oracle_lib as oracle  # noqa: E402
import randomized as R  # noqa: E402


def file_gather(d, name, rank, size, blob, timeout=120):
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


def main():
    rank, size, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    slot = int(os.environ.get("PEER_SLOT_DOUBLES", "4096"))
    rep = int(os.environ.get("PEER_REPEAT", "1"))  # soaks: every loop below this many times longer
    os.environ.setdefault("CEDR_HIP_PEER_GIVE_UP_LOG2", "25")  # a wait that is never answered ends within a minute (NaN)
    if os.environ.get("PEER_DEVICE_PER_RANK") == "1":  # a box with several GPUs: rank r on GPU r (never run so far)
        import torch
        torch.cuda.set_device(rank)
    bad = []
    try:
        mb, handle = ca.Parallel.peer_mailbox(size, slot)
        err = b""
    except ca.CedrError as e:
        mb, handle, err = None, b"\0" * ca.Parallel.PEER_HANDLE_BYTES, str(e).encode()
    errs = file_gather(d, "err", rank, size, err)
    if any(errs):  # every rank sees the same list: all leave together, before any collective
        print(json.dumps({"rank": rank, "setup_error": [e.decode() for e in errs]}))
        return 2
    handles = file_gather(d, "handle", rank, size, handle)
    par = ca.Parallel.peer_flags(mb, handles, rank, size)
    assert par.rank() == rank and par.size() == size
    file_gather(d, "mapped", rank, size, b"1")  # nobody stores into a mailbox before everybody has mapped them all

    if os.environ.get("PEER_MODE") == "giveup":
        # A peer that never arrives: rank 0's one-launch collective gives up after CEDR_HIP_PEER_GIVE_UP_LOG2 polls,
        # every wave ends, and the result is NaN (never a plausible number); the other ranks only keep their mailboxes.
        if rank == 0:
            sd = ca.DeviceArray.from_numpy(np.arange(8, dtype=np.float64))
            rd_ = ca.DeviceArray.from_numpy(np.zeros(8))
            t0 = time.monotonic()
            par.allreduce_sum(sd, rd_, 8)
            ca.synchronize()
            got = rd_.numpy()
            print(json.dumps({"rank": 0, "gave_up_nan": bool(np.isnan(got).all()), "seconds": time.monotonic() - t0,
                              "bad": [] if np.isnan(got).all() else ["a result without the peer"]}))
        else:
            print(json.dumps({"rank": rank, "bad": []}))
        file_gather(d, "done", rank, size, b"1")
        return 0

    # ---- QLT: all-gather of the ranks' subtree roots + the replicated top, twice two sets of values in turn
    ncells = int(os.environ.get("PEER_NCELLS", "1531"))
    trs = R.tracers_vector()
    types = [t.problem_type for t in trs]
    vs = [R.make_values(trs, ncells, seed=5 + s) for s in range(2)]
    wants = [common.run_oracle_qlt(oracle, (False, False), ncells, 1, types, v, False) for v in vs]
    tree = ca.Tree.make_1d(ncells, size, False, False)
    q = ca.QLT(par, ncells, tree, False)
    for t in types:
        q.declare_tracer(t, 0)
    q.end_tracer_declarations()
    q.finish_setup()
    g = q.get_owned_glblcells()
    dv = [[ca.DeviceArray.from_numpy(np.ascontiguousarray(a)) for a in
           (v.rhom[g], v.Qm[:, g], v.Qm_min[:, g], v.Qm_max[:, g], v.Qm_prev[:, g])] for v in vs]
    od = ca.DeviceArray(len(types) * len(g))
    for it in range(6 * rep):
        k = it % 2
        q.set_rhom(dv[k][0])
        q.set_Qm(dv[k][1], dv[k][2], dv[k][3], dv[k][4])
        q.run()
        q.get_Qm(od)
        ca.synchronize()
        if not np.array_equal(od.numpy((len(types), len(g))), wants[k][:, g]):
            bad.append("QLT pass %d" % it)

    # ---- CAAS: its own all-reduce (sums in rank order)
    ctrs = R.caas_tracers(trs)
    ctypes_ = [t.problem_type for t in ctrs]
    cv = R.make_values(ctrs, ncells, seed=77)
    bounds = [ncells * r // size for r in range(size + 1)]
    nl = [bounds[r + 1] - bounds[r] for r in range(size)]
    cwant, _ = common.run_oracle_caas(oracle, nl, ctypes_, cv)
    sl = slice(bounds[rank], bounds[rank + 1])
    c = ca.CAAS(par, nl[rank])
    for t in ctypes_:
        c.declare_tracer(t, 0)
    c.end_tracer_declarations()
    c.finish_setup()
    cd = [ca.DeviceArray.from_numpy(np.ascontiguousarray(a)) for a in
          (cv.rhom[sl], cv.Qm[:, sl], cv.Qm_min[:, sl], cv.Qm_max[:, sl], cv.Qm_prev[:, sl])]
    cod = ca.DeviceArray(len(ctypes_) * nl[rank])
    caas_bits = None
    for it in range(3):
        c.set_rhom(cd[0])
        c.set_Qm(cd[1], cd[2], cd[3], cd[4])
        c.run()
        c.get_Qm(cod)
        ca.synchronize()
        got = cod.numpy((len(ctypes_), nl[rank]))
        if common.rel_per_tracer(got, cwant[:, sl]).max() > 1e-13:
            bad.append("CAAS pass %d" % it)
        if caas_bits is not None and not np.array_equal(got, caas_bits):
            bad.append("CAAS pass %d differs from pass 0" % it)
        caas_bits = got.copy()

    # ---- the BFB reducer (all-gather of the ranks' roots): inputs scaled by 2^i from call to call
    nleaf, nfield = ncells, 7
    glob = np.random.default_rng(1234).random((nleaf, nfield))
    bwant = oracle.bfb_allreduce(oracle.Tree.make_1d(nleaf, 1, False, False), nleaf, 1, nfield, True,
                                 [np.ascontiguousarray(glob.T).ravel()])[0]
    btree = ca.Tree.make_1d(nleaf, size, False, False)
    ns = btree.analyze(nleaf, rank)
    mine = np.array([ns["nodes"][i][1] for i in ns["levels"][0]["nodes"]], dtype=np.int64)
    b = ca.BfbTreeAllReducer(par, btree, nleaf, nfield)
    rd = ca.DeviceArray(nfield)
    sends = [ca.DeviceArray.from_numpy(np.ascontiguousarray((glob[mine] * 2.0 ** i).T).ravel()) for i in range(4)]
    for it in range(12 * rep):
        i = it % 4
        b.allreduce(sends[i], rd, True)
        ca.synchronize()
        if not np.array_equal(rd.numpy(), bwant * 2.0 ** i):
            bad.append("BFB call %d" % it)

    # ---- a burst: 16 reducer calls enqueued back to back, no host synchronisation in between (a fast rank runs as far
    # ahead of a slow one as the transport lets it: the reuse of the two parities' slots is on its own)
    rds = [ca.DeviceArray(nfield) for _ in range(16)]
    for it in range(2 * rep):
        for j, r_ in enumerate(rds):
            b.allreduce(sends[(it + j) % 4], r_, True)
        ca.synchronize()
        for j, r_ in enumerate(rds):
            if not np.array_equal(r_.numpy(), bwant * 2.0 ** ((it + j) % 4)):
                bad.append("BFB burst %d call %d" % (it, j))

    # ---- point-to-point rounds (Comm::exchange over the mailboxes: per-pair slots, flags, acks): QLT from partial trees,
    # from a full tree with per-level rounds forced, and with pseudorandom ownership; the BFB reducer from a partial tree
    if size > 1:
        for partial, exch, pseudo in ((True, None, False), (False, "2", False), (False, None, True)):
            if exch:
                os.environ["CEDR_HIP_EXCHANGE"] = exch
            else:
                os.environ.pop("CEDR_HIP_EXCHANGE", None)
            t2 = ca.Tree.make_1d(ncells, size, pseudo, False)
            if partial:
                t2 = t2.prune(rank)
            q2 = ca.QLT(par, ncells, t2, False)
            for t in types:
                q2.declare_tracer(t, 0)
            q2.end_tracer_declarations()
            q2.finish_setup()
            if partial or exch:
                assert q2.plan()["level_mode"]
            g2 = q2.get_owned_glblcells()
            od2 = ca.DeviceArray(len(types) * len(g2))
            for it in range(2 * rep):
                v = vs[it % 2]
                d2 = [ca.DeviceArray.from_numpy(np.ascontiguousarray(a)) for a in
                      (v.rhom[g2], v.Qm[:, g2], v.Qm_min[:, g2], v.Qm_max[:, g2], v.Qm_prev[:, g2])]
                q2.set_rhom(d2[0])
                q2.set_Qm(d2[1], d2[2], d2[3], d2[4])
                q2.run()
                q2.get_Qm(od2)
                ca.synchronize()
                if not np.array_equal(od2.numpy((len(types), len(g2))), wants[it % 2][:, g2]):
                    bad.append("QLT rounds (partial %s, exchange %s, pseudorandom %s) pass %d" % (partial, exch, pseudo, it))
        os.environ.pop("CEDR_HIP_EXCHANGE", None)
        bt2 = btree.prune(rank)
        ns2 = bt2.analyze(nleaf, rank)
        mine2 = np.array([ns2["nodes"][i][1] for i in ns2["levels"][0]["nodes"]], dtype=np.int64)
        sends2 = [ca.DeviceArray.from_numpy(np.ascontiguousarray((glob[mine2] * 2.0 ** i).T).ravel()) for i in range(4)]
        b2 = ca.BfbTreeAllReducer(par, bt2, nleaf, nfield)
        for it in range(4 * rep):
            b2.allreduce(sends2[it % 4], rd, True)
            ca.synchronize()
            if not np.array_equal(rd.numpy(), bwant * 2.0 ** (it % 4)):
                bad.append("BFB from a partial tree, call %d" % it)

    # ---- QLT and CAAS in turn on two streams, one communicator (the bench's default mode): the cross-stream ordering
    import torch  # (only for the two streams)
    ts = (torch.cuda.Stream(), torch.cuda.Stream())
    q.set_stream(ts[0].cuda_stream)
    c.set_stream(ts[1].cuda_stream)
    for it in range(4 * rep):
        k = it % 2
        q.set_rhom(dv[k][0])
        q.set_Qm(dv[k][1], dv[k][2], dv[k][3], dv[k][4])
        c.set_rhom(cd[0])
        c.set_Qm(cd[1], cd[2], cd[3], cd[4])
        q.run()
        c.run()
        q.get_Qm(od)
        c.get_Qm(cod)
        ca.synchronize()
        if not np.array_equal(od.numpy((len(types), len(g))), wants[k][:, g]):
            bad.append("two streams: QLT pass %d" % it)
        if not np.array_equal(cod.numpy((len(ctypes_), nl[rank])), caas_bits):
            bad.append("two streams: CAAS pass %d" % it)

    # every rank's CAAS result rows must be what the others computed from the same sums: compare the bits of the sums'
    # effect through a digest of the global result
    digs = file_gather(d, "caas", rank, size, caas_bits.tobytes())
    allres = np.concatenate([np.frombuffer(x, dtype=np.float64).reshape(len(ctypes_), -1) for x in digs], axis=1)
    if common.rel_per_tracer(allres, cwant).max() > 1e-13:
        bad.append("CAAS global")
    print(json.dumps({"rank": rank, "size": size, "ncells": ncells, "slot_doubles": slot, "repeat": rep,
                      "collectives": 6 * rep + 3 + 12 * rep + 32 * rep + 8 * rep, "bad": bad[:20], "nbad": len(bad)}))
    file_gather(d, "done", rank, size, b"1")  # mailboxes stay mapped until every rank is through
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
