"""Spawns 2 processes (gloo) that run QLT through the product's host plan + the numpy
plan executor, exchanging subtree roots with torch.distributed.all_gather (or, from partial
trees, tree levels with send / recv); every rank
checks its cells bit-for-bit against the single-rank oracle. Test infrastructure."""
import os
import sys
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, initfile, nc, pseudorandom, use_apex, partial=False):
    os.environ["CEDR_HIP_TILE_LEAVES"] = "8"
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    import compose_amd as ca
    import common
    import oracle_lib as o
    import plan_exec
    import randomized as R
    trs = R.tracers_vector()[:12]
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, nc, seed=99)
    want = common.run_oracle_qlt(o, (False, False), nc, 1, types, v)
    bad = lambda *a: 1
    tree = ca.Tree.make_1d(nc, world, pseudorandom, False)
    if partial:  # every rank passes only its part of the tree -> per-level send / recv
        tree = tree.prune(rank)
    q = ca.QLT(ca.Parallel.callbacks_p2p(rank, world, bad, bad, bad), nc, tree)
    for t in types:
        q.declare_tracer(t, 0)
    q.end_tracer_declarations()
    st = plan_exec.RankState(q.plan(), q.metadata())
    g = q.get_owned_glblcells()
    st.set(v.rhom[g], v.Qm[:, g], v.Qm_min[:, g], v.Qm_max[:, g], v.Qm_prev[:, g])
    st.up_local()
    if partial:
        assert st.plan["level_mode"]
        st.owns_root = st.plan["root_owner"] == rank

        def xch(tag, sends, recvs):
            reqs = [dist.isend(torch.from_numpy(np.ascontiguousarray(a)), peer) for peer, a in sends]
            got = []
            for peer, cnt in recvs:
                t = torch.zeros(cnt, st.l2rndps if tag[0] == "up" else st.r2lndps, dtype=torch.float64)
                dist.recv(t, peer)
                got.append(t.numpy())
            for r in reqs:
                r.wait()
            return got
        st.run_level_mode(xch)
    else:
        send = torch.from_numpy(st.send_records())
        bufs = [torch.zeros_like(send) for _ in range(world)]
        dist.all_gather(bufs, send)
        st.recv_records([b.numpy() for b in bufs])
        assert not use_apex or st.plan["apex"] is not None
        st.top_and_down(use_apex)  # the top as the product's default (one apex tile) or staged
    ok = np.array_equal(st.get(len(g)), want[:, g])
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    if not int(t[0]):
        sys.exit(1)


if __name__ == "__main__":
    for nc, pr, apex in ((24, False, True), (25, True, True), (25, True, False)):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(worker, args=(2, os.path.join(d, "init"), nc, pr, apex), nprocs=2, join=True)
    for nc, pr in ((24, False), (40, True)):  # partial trees: level mode over send / recv
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(worker, args=(2, os.path.join(d, "init"), nc, pr, False, True), nprocs=2, join=True)
    print("GLOO_OK")
