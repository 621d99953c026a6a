"""Parity at the sizes BASELINE.json quotes. Config B (150k x 10) is checked against the
oracle directly; config C (1.5M x 40) through the size-independent properties of
cedr_test_randomized.cpp:293-418 (bounds, mass conservation) plus a checksum comparison
with the oracle on a strided sample of tracers; a 1M-cell x 600-tracer problem puts field
offsets beyond 2^31 elements (the reference's Int indexing overflows there)."""
import ctypes as C
import os

import numpy as np
import pytest

import common
import randomized as R

pytestmark = pytest.mark.gpu
CST = 7


def _inputs(n, nt, seed):
    rng = np.random.default_rng(seed)
    rhom = 0.5 * (1 + rng.random(n))
    q_min = -0.75 + rng.random((nt, n))
    q_max = q_min + rng.random((nt, n))
    q = q_min + (q_max - q_min) * rng.random((nt, n))
    v = R.Values(nt, n)
    v.rhom[:] = rhom
    v.Qm_min[:], v.Qm_max[:] = q_min * rhom, q_max * rhom
    v.Qm_prev[:] = np.maximum(v.Qm_min, np.minimum(v.Qm_max, q * rhom))
    for t in range(nt):  # perturbation 1: permute
        v.Qm[t] = v.Qm_prev[t][rng.permutation(n)]
    return v


def test_config_B_vs_oracle(gpu, oracle):
    ca = gpu
    n, nt = 150000, 10
    v = _inputs(n, nt, 1)
    q = common.make_hip_qlt(ca, ca.Tree.make_1d(n), n, [CST] * nt)
    got = common.run_hip_qlt(ca, q, v)
    want = common.run_oracle_qlt(oracle, (False, False), n, 1, [CST] * nt, v)
    assert np.array_equal(got, want)
    cg = common.run_hip_caas(ca, n, [CST] * nt, v)
    cw, _ = common.run_oracle_caas(oracle, [n], [CST] * nt, v)
    assert common.rel_per_tracer(cg, cw).max() <= 1e-13


def test_config_C_properties(gpu, oracle):
    ca = gpu
    n, nt = 1500000, 40
    v = _inputs(n, nt, 2)
    q = common.make_hip_qlt(ca, ca.Tree.make_1d(n), n, [CST] * nt)
    got = common.run_hip_qlt(ca, q, v)
    assert (got >= v.Qm_min).all() and (got <= v.Qm_max).all()
    rel = np.abs(got.sum(1) - v.Qm_prev.sum(1)) / np.abs(v.Qm_prev).sum(1)
    assert rel.max() <= 1e-13
    # three tracers against the oracle (it runs any tracer subset: tracers are independent)
    sub = [0, 17, 39]
    vs = R.Values(len(sub), n)
    vs.rhom[:] = v.rhom
    for k, t in enumerate(sub):
        vs.Qm[k], vs.Qm_min[k], vs.Qm_max[k], vs.Qm_prev[k] = v.Qm[t], v.Qm_min[t], v.Qm_max[t], v.Qm_prev[t]
    want = common.run_oracle_qlt(oracle, (False, False), n, 1, [CST] * len(sub), vs)
    assert np.array_equal(got[sub], want)
    cg = common.run_hip_caas(ca, n, [CST] * nt, v)
    assert (cg >= v.Qm_min).all() and (cg <= v.Qm_max).all()
    rel = np.abs(cg.sum(1) - v.Qm_prev.sum(1)) / np.abs(v.Qm_prev).sum(1)
    assert rel.max() <= 1e-13
    # CAAS at the full size against (1) the oracle = the reference's sequential double sums and
    # (2) the same formula (cedr_caas.cpp:129-253) with EXACTLY rounded sums (math.fsum): SURVEY 7
    # flagged n = 1.5M as the size where a parallel sum's rounding approaches the 1e-13 tolerance.
    # Per-tracer relative norms; the margins are printed (pytest -s) and recorded in DESIGN.md.
    cw, _ = common.run_oracle_caas(oracle, [n], [CST] * nt, v)
    r_oracle = common.rel_per_tracer(cg, cw)
    few = [0, 7, 13, 21, 30, 39]
    exact = common.caas_exact_sums(v, few)
    r_exact = common.rel_per_tracer(cg[few], exact)
    r_oracle_exact = common.rel_per_tracer(cw[few], exact)
    print("CAAS 1.5M x 40: HIP vs oracle max %.2e; HIP vs exact sums max %.2e; oracle vs exact sums max %.2e"
          % (r_oracle.max(), r_exact.max(), r_oracle_exact.max()))
    assert r_oracle.max() <= 1e-13 and r_exact.max() <= 1e-13


def test_offsets_beyond_int32(gpu, oracle):
    """1M cells x 600 identical cst tracers: l2r has 2401 field rows of > 2M slots, i.e. field
    offsets up to 4.9e9 elements. Every tracer must reproduce tracer 0's oracle result."""
    ca = gpu
    L = ca.lib()
    n, nt = 1000000, 600
    v = _inputs(n, 1, 3)
    want = common.run_oracle_qlt(oracle, (False, False), n, 1, [CST], v)[0]
    q = common.make_hip_qlt(ca, ca.Tree.make_1d(n), n, [CST] * nt)
    b1, _ = q.get_buffers_sizes()
    assert b1 > 2**31
    rows = {k: ca.DeviceArray.from_numpy(a[0]) for k, a in
            (("Qm", v.Qm), ("lo", v.Qm_min), ("hi", v.Qm_max), ("prev", v.Qm_prev))}
    big = {k: ca.DeviceArray(n * nt) for k in rows}
    for k in rows:
        for t in range(nt):
            ca._chk(L.cedr_memcpy_d2d_async(big[k].offset(t * n), rows[k].ptr, 8 * n, None))
    q.set_rhom(ca.DeviceArray.from_numpy(v.rhom))
    q.set_Qm(big["Qm"], big["lo"], big["hi"], big["prev"])
    q.run()
    out = ca.DeviceArray(n * nt)
    q.get_Qm(out)
    ca.synchronize()
    row = np.empty(n)
    for t in (0, 1, 299, 598, 599):
        ca._chk(L.cedr_memcpy_d2h(row.ctypes.data, out.offset(t * n), row.nbytes))
        assert np.array_equal(row, want), t


def test_config_D_sharded_8_ways(gpu, oracle):
    """BASELINE config D: 6M cells x 128 cst tracers, sharded 8 ways. The 8 ranks run as 8
    threads on this one GPU (tests/threaded_ranks.py: the real gather / all-gather / scatter /
    replicated-top path, host-staged transport) and must reproduce the single-rank run of the
    same problem bit for bit; that run is held to the harness's properties, and three of its
    tracers to the oracle (tracers are independent, so the oracle runs just those)."""
    import os
    import sys
    import torch
    import threaded_ranks as T
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    ca = gpu
    n, nt, P = 6000000, 128, 8
    dev = torch.device("cuda", 0)
    rhom, Qm, lo, hi, prev = bench.gen_inputs(torch, n, nt, 0x5EED0000 + 4000, dev)
    torch.cuda.synchronize()

    q1 = common.make_hip_qlt(ca, ca.Tree.make_1d(n), n, [CST] * nt)
    b1, b2 = q1.get_buffers_sizes()
    assert b1 == (2 * n - 1) * (1 + 4 * nt) and b2 == (2 * n - 1) * nt  # reference's sizes (doubles)
    q1.set_rhom(rhom)
    q1.set_Qm(Qm, lo, hi, prev)
    q1.run()
    out1 = torch.empty_like(Qm)
    q1.get_Qm(out1)
    ca.synchronize()
    del q1
    assert bool(((out1 >= lo) & (out1 <= hi)).all())
    rel = (out1.sum(1) - prev.sum(1)).abs() / prev.abs().sum(1)
    assert float(rel.max()) <= 1e-13

    out8 = torch.zeros_like(Qm)

    def body(rank, comm):
        q = ca.QLT(comm.parallel(ca, rank), n, ca.Tree.make_1d(n, P, False, False))
        for _ in range(nt):
            q.declare_tracer(CST, 0)
        q.end_tracer_declarations()
        q.finish_setup()
        g = torch.from_numpy(q.get_owned_glblcells().astype(np.int64)).to(dev)
        loc = [x.index_select(x.dim() - 1, g).contiguous() for x in (rhom, Qm, lo, hi, prev)]
        torch.cuda.synchronize()
        q.set_rhom(loc[0])
        q.set_Qm(loc[1], loc[2], loc[3], loc[4])
        q.run()
        o = torch.empty_like(loc[1])
        q.get_Qm(o)
        ca.synchronize()
        out8[:, g] = o
        return int(g.numel())
    counts = T.run_ranks(P, body, timeout=1200)
    assert counts == [n // P] * P
    torch.cuda.synchronize()
    assert torch.equal(out1, out8)

    sub = [0, 63, 127]
    vs = R.Values(len(sub), n)
    vs.rhom[:] = rhom.cpu().numpy()
    for k, t in enumerate(sub):
        vs.Qm[k], vs.Qm_min[k], vs.Qm_max[k], vs.Qm_prev[k] = (x[t].cpu().numpy() for x in (Qm, lo, hi, prev))
    want = common.run_oracle_qlt(oracle, (False, False), n, 1, [CST] * len(sub), vs)
    assert np.array_equal(out1[sub].cpu().numpy(), want)


def test_config_C_sharded_8_ways(gpu, oracle):
    """BASELINE config 4: QLT + CAAS, 1.5M cells x 40 cst tracers, sharded 8 ways -- the 8 ranks as 8
    threads on this one GPU (tests/threaded_ranks.py: the real gather / all-gather / scatter /
    replicated-top path and the CAAS all-reduce, host-staged transport).
    QLT: bit-identical to the single-rank run; three tracers bit-identical to the oracle.
    CAAS with the BFB tree all-reduce as its UserAllReducer (the production pairing), fused (the reducer object is
    the UserAllReducer) and as two launches (a callback): bit-identical to each other and on 1 and 8 ranks. CAAS with its own all-reduce: per tracer within 1e-13 of the single-rank run and of
    the oracle (8 partial sums are added in another order than one rank's)."""
    import os
    import sys
    import torch
    import threaded_ranks as T
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    ca = gpu
    n, nt, P, acc = 1500000, 40, 8, 4
    dev = torch.device("cuda", 0)
    rhom, Qm, lo, hi, prev = bench.gen_inputs(torch, n, nt, 0x5EED0000 + 3000, dev)
    torch.cuda.synchronize()
    glob = (rhom, Qm, lo, hi, prev)

    def declare(c):
        for _ in range(nt):
            c.declare_tracer(CST, 0)
        c.end_tracer_declarations()
        c.finish_setup()
        return c

    def run_all(nranks, partial=False):
        """(QLT, CAAS own all-reduce, CAAS + BFB reducer) results as global [nt][n] tensors.
        partial: QLT only, every rank from its own partial tree (tree::Node::level -> level mode)."""
        outs = [torch.zeros_like(Qm) for _ in range(4)]

        def body(rank, comm):
            par = comm.parallel(ca, rank)
            tree = ca.Tree.make_1d(n, nranks, False, False)
            if partial:
                tree = tree.prune(rank)
                assert tree.nnodes() < 2 * (n // nranks) + 64
            q = declare(ca.QLT(par, n, tree))
            # (CEDR_HIP_EXCHANGE=2, one of the env-matrix settings, forces level mode for full trees too)
            assert q.plan()["level_mode"] == (partial or (nranks > 1 and os.environ.get("CEDR_HIP_EXCHANGE") == "2"))
            g = torch.from_numpy(q.get_owned_glblcells().astype(np.int64)).to(dev)
            assert int(g[0]) == rank * (n // nranks) and bool((g[1:] - g[:-1] == 1).all())  # contiguous shard
            loc = [x.index_select(x.dim() - 1, g).contiguous() for x in glob]
            torch.cuda.synchronize()
            q.set_rhom(loc[0]); q.set_Qm(loc[1], loc[2], loc[3], loc[4]); q.run()
            o = torch.empty_like(loc[1])
            q.get_Qm(o); ca.synchronize()
            outs[0][:, g] = o
            del q
            if partial:
                return int(g.numel())
            c = declare(ca.CAAS(par, len(g)))
            c.set_rhom(loc[0]); c.set_Qm(loc[1], loc[2], loc[3], loc[4]); c.run()
            c.get_Qm(o); ca.synchronize()
            outs[1][:, g] = o
            del c
            btree = ca.Tree.make_1d(n // acc, nranks, False, False)
            bfb = ca.BfbTreeAllReducer(par, btree, n // acc, 4 * nt)
            assert bfb.nlocal() == len(g) // acc
            # the reducer object itself as the UserAllReducer (caas::BfbUserAllReducer): run() fuses the reduce pass
            # with the first stage of the reducer's tree
            cb = declare(ca.CAAS(par, len(g), bfb, acc))
            assert cb.fused_with_bfb() or not common.default_knobs()
            cb.set_rhom(loc[0]); cb.set_Qm(loc[1], loc[2], loc[3], loc[4]); cb.run()
            cb.get_Qm(o); ca.synchronize()
            outs[2][:, g] = o
            del cb
            # ... and through a callback: the two separate launches
            cb = declare(ca.CAAS(par, len(g), lambda s, r, nl, nf: bfb.allreduce(s, r, True), acc))
            assert not cb.fused_with_bfb()
            cb.set_rhom(loc[0]); cb.set_Qm(loc[1], loc[2], loc[3], loc[4]); cb.run()
            cb.get_Qm(o); ca.synchronize()
            outs[3][:, g] = o
            return int(g.numel())
        counts = T.run_ranks(nranks, body, timeout=1200)
        assert counts == [n // nranks] * nranks
        torch.cuda.synchronize()
        return outs

    q1, c1, b1, u1 = run_all(1)
    q8, c8, b8, u8 = run_all(P)
    assert torch.equal(b1, u1) and torch.equal(b8, u8), "the fused CAAS + BFB path differs from the separate launches"
    assert torch.equal(q1, q8), "QLT depends on the rank count"
    assert torch.equal(q1, run_all(P, partial=True)[0]), "QLT from partial trees (level mode) differs"
    assert torch.equal(b1, b8), "CAAS + BFB reducer depends on the rank count"
    rel = lambda a, b: float(((a - b).abs().amax(1) / b.abs().amax(1)).max())
    assert rel(c8, c1) <= 1e-13 and rel(b1, c1) <= 1e-13
    for out in (q1, c1, b1):
        assert bool(((out >= lo) & (out <= hi)).all())
        assert float(((out.sum(1) - prev.sum(1)).abs() / prev.abs().sum(1)).max()) <= 1e-13
    sub = [0, 19, 39]
    vs = R.Values(len(sub), n)
    vs.rhom[:] = rhom.cpu().numpy()
    for k, t in enumerate(sub):
        vs.Qm[k], vs.Qm_min[k], vs.Qm_max[k], vs.Qm_prev[k] = (x[t].cpu().numpy() for x in (Qm, lo, hi, prev))
    want = common.run_oracle_qlt(oracle, (False, False), n, 1, [CST] * len(sub), vs)
    assert np.array_equal(q8[sub].cpu().numpy(), want)
    cw, _ = common.run_oracle_caas(oracle, [n], [CST] * len(sub), vs)
    for got in (c8, b8):
        assert common.rel_per_tracer(got[sub].cpu().numpy(), cw).max() <= 1e-13


def test_qlt_placement_tries_do_not_change_results(gpu, monkeypatch):
    """QLT::finish_setup tries several allocations of its l2r buffer (CEDR_HIP_PLACE_TRIES) and times the first-stage
    sweeps over them; the buffers it ends up with are clean and the results are the same bits as with one try."""
    ca = gpu
    ncells, nt = 400000, 3  # > 2048 first-stage tiles: enough for the tries to run
    trs = R.tracers_vector()[:nt]
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, ncells, seed=12)
    outs = []
    for tries in ("1", "3"):
        monkeypatch.setenv("CEDR_HIP_PLACE_TRIES", tries)
        q = common.make_hip_qlt(ca, ca.Tree.make_1d(ncells), ncells, types)
        assert len(q.plan()["local"][0]["tiles"]) >= 2048
        outs.append(common.run_hip_qlt(ca, q, v))
    assert np.array_equal(outs[0], outs[1])
    assert R.check(trs, v, outs[1], False, verbose=False)[0] == 0


def test_tail_split_does_not_change_results(gpu, oracle, monkeypatch):
    """The first-stage launches end with short workgroups: the tiles past an XCD's last full round of workgroups are
    dealt to several workgroups by chunks (CEDR_HIP_TAIL, CEDR_HIP_TAIL_ROUNDS). Same chunks, other workgroups: the
    results are the oracle's bits whatever the split, for shape-preserving tracers (the specialised root->leaves
    kernel) and for the harness's mixed types (the general one, which is launched unsplit)."""
    ca = gpu
    ncells = 600000  # 4096 first-stage tiles: three full rounds per XCD and a tail
    cst = [R.Tracer(i, CST, 1 + i % 3) for i in range(12)]  # three chunks of four; perturbations that redistribute
    for trs in (cst, R.tracers_vector()[:9]):
        types = [t.problem_type for t in trs]
        v = R.make_values(trs, ncells, seed=len(trs))
        want = common.run_oracle_qlt(oracle, (False, False), ncells, 1, types, v)
        for tail, rounds in (("0", "0"), ("2", "0"), ("5", "0"), ("3", "1"), ("64", "2")):
            monkeypatch.setenv("CEDR_HIP_TAIL", tail)
            monkeypatch.setenv("CEDR_HIP_TAIL_ROUNDS", rounds)
            q = common.make_hip_qlt(ca, ca.Tree.make_1d(ncells), ncells, types)
            assert len(q.plan()["local"][0]["tiles"]) >= 4096
            got = common.run_hip_qlt(ca, q, v)
            assert np.array_equal(got, want), (len(trs), tail, rounds)
            del q


def test_mixed_types_class_forms_at_scale(gpu, oracle, monkeypatch):
    """A mixed tracer list at a size whose first stage has >= 2048 tiles: the root->leaves launch over the cells is swept
    class by class with the class-specialised forms of the shape-preserving kernel (shape preserving / t-only /
    nonnegative; cedr_qlt.cpp:490-523,549-560, cedr_qlt_inl.hpp:190-198), the smaller stages by the general kernel:
    bit-identical to the oracle, to the general kernel alone (CEDR_HIP_DOWN_CLS=0), and with the cells' bound rows of
    t-only tracers written as the reference leaves them (CEDR_HIP_LEAF_BOUNDS=1)."""
    ca = gpu
    n = 400000
    trs = R.tracers_vector()[:12]  # all six problem types, two perturbations
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, n, seed=2026)
    want = common.run_oracle_qlt(oracle, (False, False), n, 1, types, v)
    outs = []
    for env in ({}, {"CEDR_HIP_DOWN_CLS": "0"}, {"CEDR_HIP_LEAF_BOUNDS": "1"}):
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        q = common.make_hip_qlt(ca, ca.Tree.make_1d(n), n, types)
        if not env and common.default_knobs():
            assert len(q.plan()["local"][0]["tiles"]) >= 2048
        outs.append(common.run_hip_qlt(ca, q, v))
        for k in env:
            monkeypatch.delenv(k)
        del q
    for got in outs:
        assert np.array_equal(got, want)
    assert R.check(trs, v, outs[0], False, verbose=False)[0] == 0


def test_set_stream_orders_the_new_stream_after_the_old(gpu):
    """set_Qm on one stream (here the null stream), then set_stream(a non-blocking stream) and run() at once, no host
    synchronisation in between: the new stream waits for what the object enqueued on the old one (a size at which the
    packing kernels are still running when run() is enqueued -- without the ordering its first result was garbage)."""
    import sys
    import torch
    ca = gpu
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n, nt = 750000, 40
    dev = torch.device("cuda", 0)
    rhom, Qm, lo, hi, prev = bench.gen_inputs(torch, n, nt, 77, dev)
    par = ca.Parallel.self_()
    q = ca.QLT(par, n, ca.Tree.make_1d(n, 1, False, False))
    c = ca.CAAS(par, n)
    for x in (q, c):
        for _ in range(nt):
            x.declare_tracer(7, 0)
        x.end_tracer_declarations()
        x.finish_setup()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(device=dev)
    for x in (q, c):
        x.set_stream(None)
        x.set_rhom(rhom)
        x.set_Qm(Qm, lo, hi, prev)
        x.set_stream(s.cuda_stream)
        x.run()
        first = torch.empty_like(Qm)
        x.get_Qm(first)
        torch.cuda.synchronize()
        x.run()
        again = torch.empty_like(Qm)
        x.get_Qm(again)
        torch.cuda.synchronize()
        if x is q:  # (CAAS::run without set_Qm_source clips in place: its second run starts from the first one's result)
            assert torch.equal(first, again)
        x.set_stream(None)
        x.set_rhom(rhom)
        x.set_Qm(Qm, lo, hi, prev)
        torch.cuda.synchronize()
        x.run()
        ref = torch.empty_like(Qm)
        x.get_Qm(ref)
        torch.cuda.synchronize()
        assert torch.equal(first, ref), type(x).__name__
