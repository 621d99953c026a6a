"""Multi-rank data path on one GPU: P ranks as P threads (tests/threaded_ranks.py),
checked bit-for-bit against the single-rank oracle -- QLT results may not depend on
the rank count or on who owns which cell (reference cedr_qlt.cpp:745-772 runs the same
harness on any communicator size)."""
import numpy as np
import pytest

import common
import randomized as R
import threaded_ranks as T


def _bfb_local_leaves(tree, nleaf, rank):
    """Global ids of a rank's leaves in the order BfbTreeAllReducer indexes its send buffer:
    NodeSets levels[0].nodes (cedr_bfb_tree_allreduce.cpp:91-102) -- which is NOT always
    QLT's local cell order (node offsets) when ownership cuts across subtrees."""
    ns = tree.analyze(nleaf, rank)
    return np.array([ns["nodes"][i][1] for i in ns["levels"][0]["nodes"]], dtype=np.int64)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nranks,ncells,pseudorandom,imbalanced", [
    (2, 24, False, False), (3, 111, False, True), (4, 1000, False, False), (8, 1000, True, False),
    (8, 5000, False, False), (8, 8, False, False), (5, 1024, True, True)])
@pytest.mark.parametrize("prefer", [False, True])
def test_qlt_ranks_as_threads_bitwise(gpu, oracle, nranks, ncells, pseudorandom, imbalanced, prefer):
    trs = R.tracers_vector()
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, ncells, seed=4242 + ncells)
    want = common.run_oracle_qlt(oracle, (False, imbalanced), ncells, 1, types, v, prefer)
    got = T.run_qlt(gpu, nranks, ncells, types, v, prefer, pseudorandom, imbalanced, nrepeat=2)
    assert np.array_equal(got, want)
    assert R.check(trs, v, got, prefer, verbose=False)[0] == 0


@pytest.mark.parametrize("nranks,ncells", [(2, 1000), (8, 5000), (8, 4096)])
def test_caas_ranks_as_threads(gpu, oracle, nranks, ncells):
    trs = R.caas_tracers(R.tracers_vector())
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, ncells, seed=77)
    got, nlcl = T.run_caas(gpu, nranks, ncells, types, v)
    want, _ = common.run_oracle_caas(oracle, nlcl, types, v)
    assert common.rel_per_tracer(got, want).max() <= 1e-13
    assert R.check(trs, v, got, False, verbose=False)[0] == 0


@pytest.mark.parametrize("partial", [False, True])
@pytest.mark.parametrize("nranks,ncells", [(4, 3000), (3, 1000), (5, 1000), (8, 1001)])
def test_qlt_bound_arrays_ranks_as_threads(gpu, oracle, nranks, ncells, partial):
    """Zero-copy mode across ranks: every rank binds its own slices. Ownership that cuts
    through subtrees (3, 5 ranks) includes ranks that own a lone cell as a whole subtree,
    which fall back to the packed path inside run(); the result may not depend on it."""
    ca = gpu
    trs = R.tracers_vector()
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, ncells, seed=5)
    want = common.run_oracle_qlt(oracle, (False, False), ncells, 1, types, v)
    out = np.zeros_like(v.Qm)

    def body(rank, comm):
        tree = ca.Tree.make_1d(ncells, nranks, False, False)
        if partial:  # level mode: the same zero-copy rules apply to the subtrees a rank owns entirely
            tree = tree.prune(rank)
        q = ca.QLT(comm.parallel(ca, rank), ncells, tree)
        for t in types:
            q.declare_tracer(t, 0)
        q.end_tracer_declarations()
        q.finish_setup()
        g = q.get_owned_glblcells()
        d = [ca.DeviceArray.from_numpy(a) for a in
             (v.rhom[g], v.Qm[:, g], v.Qm_min[:, g], v.Qm_max[:, g], v.Qm_prev[:, g])]
        q.bind_arrays(d[0], d[1], d[2], d[3], d[4], d[1])
        q.run()
        ca.synchronize()
        out[:, g] = d[1].numpy((len(types), len(g)))
    T.run_ranks(nranks, body)
    assert np.array_equal(out, want)


@pytest.mark.parametrize("typed", [False, True])
@pytest.mark.parametrize("accum,ngroups", [(1, 96), (4, 96), (4, 24000), (3, 20011)])
def test_caas_with_bfb_user_reducer_is_decomposition_invariant(gpu, oracle, accum, ngroups, typed):
    """The production pairing (north star): CAAS whose UserAllReducer is the BFB tree
    all-reduce over element partials. The limited masses must be bit-identical on 1, 2, 4
    and 8 ranks -- that is the point of the BFB reducer -- and agree with the oracle's CAAS."""
    ca = gpu
    # ngroups: leaves of the reducer's tree ("elements"); typed: the reducer object itself is the UserAllReducer
    # (caas::BfbUserAllReducer), which fuses CAAS's reduce pass with the tree's first stage on ranks whose part of
    # the tree has one (the large cases) -- and must not change a bit
    ncells = ngroups * accum          # cells per element = n_accum_in_place
    trs = R.caas_tracers(R.tracers_vector())
    types = [t.problem_type for t in trs]
    nt = len(types)
    v = R.make_values(trs, ncells, seed=accum)
    results = {}
    for nranks in (1, 2, 4, 8):
        out = np.zeros_like(v.Qm)

        def body(rank, comm):
            par = comm.parallel(ca, rank)
            tree = ca.Tree.make_1d(ngroups, nranks, False, False)
            groups = _bfb_local_leaves(tree, ngroups, rank)
            cells = (groups[:, None] * accum + np.arange(accum)[None, :]).ravel()
            bfb = ca.BfbTreeAllReducer(par, tree, ngroups, 4 * nt)
            assert bfb.nlocal() == len(groups)

            def reducer(send, recv, nlocal, nfld):
                assert nlocal == len(groups) and nfld == 4 * nt
                bfb.allreduce(send, recv, True)  # CAAS emits (field, partial): the transposed layout
                return 0
            c = ca.CAAS(par, len(cells), bfb if typed else reducer, accum)
            for t in types:
                c.declare_tracer(t, 0)
            c.end_tracer_declarations()
            c.finish_setup()
            if typed and ngroups == 24000 and common.default_knobs():  # (24000 halves three times into whole subtrees: every rank's part is fusable;
                assert c.fused_with_bfb()   # of 20011 leaves some ranks own a lone leaf as a subtree of its own and are not)
            if not typed:
                assert not c.fused_with_bfb()
            d = [ca.DeviceArray.from_numpy(a) for a in
                 (v.rhom[cells], v.Qm[:, cells], v.Qm_min[:, cells], v.Qm_max[:, cells], v.Qm_prev[:, cells])]
            c.set_rhom(d[0])
            c.set_Qm(d[1], d[2], d[3], d[4])
            c.run()
            od = ca.DeviceArray(nt * len(cells))
            c.get_Qm(od)
            ca.synchronize()
            out[:, cells] = od.numpy((nt, len(cells)))
        T.run_ranks(nranks, body)
        results[nranks] = out
    for nranks in (2, 4, 8):
        assert np.array_equal(results[nranks], results[1]), nranks
    want, _ = common.run_oracle_caas(oracle, [ncells], types, v)
    assert common.rel_per_tracer(results[1], want).max() <= 1e-13
    assert R.check(trs, v, results[1], False, verbose=False)[0] == 0
    assert np.array_equal(results[1], common.caas_bfb_expected(oracle, v, types, ngroups, accum))


@pytest.mark.parametrize("nranks,ncell", [(2, 24), (3, 1000), (5, 1000), (8, 1000), (8, 8)])
@pytest.mark.parametrize("transpose", [False, True])
def test_bfb_allreduce_ranks_as_threads(gpu, oracle, nranks, ncell, transpose):
    """BfbTreeAllReducer across ranks (cedr_bfb_tree_allreduce.cpp:79-159): every rank gets
    the same bits, equal to the single-rank oracle sum over the same tree."""
    ca = gpu
    nfield = 5
    rng = np.random.default_rng(ncell + nranks)
    glob = rng.random((ncell, nfield))  # [leaf][field]
    lay = lambda a: np.ascontiguousarray(a.T if transpose else a).ravel()
    want = oracle.bfb_allreduce(oracle.Tree.make_1d(ncell, 1, False, False), ncell, 1, nfield, transpose,
                                [lay(glob)])[0]
    got = [None] * nranks

    def body(rank, comm):
        par = comm.parallel(ca, rank)
        tree = ca.Tree.make_1d(ncell, nranks, False, False)
        mine = _bfb_local_leaves(tree, ncell, rank)
        b = ca.BfbTreeAllReducer(par, tree, ncell, nfield)
        assert b.nlocal() == len(mine)
        sd = ca.DeviceArray.from_numpy(lay(glob[mine]))
        rd = ca.DeviceArray(nfield)
        b.allreduce(sd, rd, transpose)
        ca.synchronize()
        got[rank] = rd.numpy()
    T.run_ranks(nranks, body)
    for r in range(nranks):
        assert np.array_equal(got[r], want), r


# ---- level mode: partial caller trees (tree::Node::level) and scattered ownership
@pytest.mark.parametrize("nranks,ncells,pseudorandom,imbalanced", [
    (2, 24, False, False), (3, 111, False, True), (4, 1000, False, False), (8, 1000, True, False),
    (8, 5000, False, False), (8, 8, False, False), (5, 1024, True, True)])
@pytest.mark.parametrize("partial", [False, True])
def test_qlt_level_mode_ranks_as_threads_bitwise(gpu, oracle, nranks, ncells, pseudorandom, imbalanced, partial,
                                                 monkeypatch):
    """Every rank passes only its part of the tree (partial), or the full tree with the per-level
    exchange forced: pure subtrees by the tile kernels, the rest level by level with point-to-point
    messages (reference cedr_qlt.cpp:618-640) -- the same bits as the single-rank oracle."""
    if not partial:
        monkeypatch.setenv("CEDR_HIP_EXCHANGE", "2")
    trs = R.tracers_vector()
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, ncells, seed=99 + ncells)
    want = common.run_oracle_qlt(oracle, (False, imbalanced), ncells, 1, types, v, False)
    got = T.run_qlt(gpu, nranks, ncells, types, v, False, pseudorandom, imbalanced, nrepeat=2, partial=partial,
                    want_level_mode=True)
    assert np.array_equal(got, want)
    assert R.check(trs, v, got, False, verbose=False)[0] == 0


def test_qlt_level_mode_larger_scattered(gpu, oracle):
    """Pseudorandom ownership at a size where the automatic rule picks level mode by itself
    (the all-gather would hand every rank the whole tree): 6 ranks, 120k cells, 3 tracers."""
    nranks, ncells = 6, 120000
    trs = R.tracers_vector()[:3]
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, ncells, seed=6)
    want = common.run_oracle_qlt(oracle, (False, False), ncells, 1, types, v, False)
    got = T.run_qlt(gpu, nranks, ncells, types, v, False, True, False, want_level_mode=True)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("nranks,ncell", [(2, 24), (3, 1000), (8, 1000), (8, 8)])
@pytest.mark.parametrize("transpose", [False, True])
def test_bfb_level_mode_ranks_as_threads(gpu, oracle, nranks, ncell, transpose):
    """BfbTreeAllReducer over partial trees: the root's owner broadcasts the result."""
    ca = gpu
    nfield = 5
    rng = np.random.default_rng(ncell + nranks)
    glob = rng.random((ncell, nfield))
    lay = lambda a: np.ascontiguousarray(a.T if transpose else a).ravel()
    want = oracle.bfb_allreduce(oracle.Tree.make_1d(ncell, 1, False, False), ncell, 1, nfield, transpose,
                                [lay(glob)])[0]
    got = [None] * nranks

    def body(rank, comm):
        par = comm.parallel(ca, rank)
        tree = ca.Tree.make_1d(ncell, nranks, False, False).prune(rank)
        mine = _bfb_local_leaves(tree, ncell, rank)
        b = ca.BfbTreeAllReducer(par, tree, ncell, nfield)
        assert b.nlocal() == len(mine)
        sd = ca.DeviceArray.from_numpy(lay(glob[mine]))
        rd = ca.DeviceArray(nfield)
        for _ in range(2):
            b.allreduce(sd, rd, transpose)
        ca.synchronize()
        got[rank] = rd.numpy()
    T.run_ranks(nranks, body)
    for r in range(nranks):
        assert np.array_equal(got[r], want), r


def test_level_mode_needs_point_to_point(gpu, monkeypatch):
    """A transport without the exchange callback (mpi::Comm::has_exchange false) runs all-gather plans
    only. A partial tree needs the per-level messages: refused when the CDR is CONSTRUCTED, on every
    rank alike, not at the first level step inside run(). A full tree with scattered ownership, which
    the planner would otherwise put in level mode, is planned for the all-gather instead."""
    ca = gpu
    bad = lambda *a: 1
    with pytest.raises(ca.CedrError, match="point-to-point"):
        ca.QLT(ca.Parallel.callbacks(0, 2, bad, bad), 64, ca.Tree.make_1d(64, 2).prune(0))
    monkeypatch.setenv("CEDR_HIP_EXCHANGE", "2")
    with pytest.raises(ca.CedrError, match="point-to-point"):
        ca.QLT(ca.Parallel.callbacks(0, 2, bad, bad), 64, ca.Tree.make_1d(64, 2))
    monkeypatch.delenv("CEDR_HIP_EXCHANGE")
    nc = 40000  # pseudorandom ownership: level mode by the planner's own rule ...
    assert ca.Tree.make_1d(nc, 4, True, False).plan(nc, 1, 4, 0)["level_mode"]
    q = ca.QLT(ca.Parallel.callbacks(1, 4, bad, bad), nc, ca.Tree.make_1d(nc, 4, True, False))
    q.declare_tracer(7, 0)
    q.end_tracer_declarations()
    q.finish_setup()
    assert not q.plan()["level_mode"]  # ... but not over a transport that cannot exchange


@pytest.mark.parametrize("kind,nleaf,nranks,how,pin", [
    ("merge", 40, 2, "blocks", 0.0), ("merge", 3001, 5, "blocks", 0.0), ("merge", 800, 4, "scattered", 0.0),
    ("skew", 1500, 3, "blocks", 0.0), ("chain", 300, 3, "blocks", 0.0), ("chain", 45, 4, "scattered", 0.0),
    ("merge", 1200, 4, "blocks", 0.3), ("skew", 900, 6, "scattered", 0.5), ("merge", 20000, 8, "blocks", 0.0)])
def test_qlt_random_caller_trees_ranks_as_threads(gpu, oracle, kind, nleaf, nranks, how, pin, monkeypatch):
    """Irregular caller trees (cedr_tree_caller.hpp:12-30) on several ranks: block and scattered
    ownership, internal nodes pinned to ranks that own nothing below them. The all-gather path
    (or what the planner picks when nodes are pinned), level mode on the full tree, and level mode
    from per-rank partial trees all give the single-rank oracle's bits."""
    import random_trees as RT
    ca = gpu
    rng = np.random.default_rng(7 * nleaf + nranks)
    root, kid0, kid1, rank, cell = RT.random_tree(rng, nleaf, kind)
    rank = RT.assign_ranks(rng, nleaf, rank, nranks, how, pin)
    trs = R.tracers_vector()
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, nleaf, seed=nleaf)
    want = RT.oracle_single_rank(oracle, root, kid0, kid1, cell, nleaf, types, v)
    for mode, partial in ((0 if pin > 0 else 1, False), (2, False), (2, True)):
        monkeypatch.setenv("CEDR_HIP_EXCHANGE", str(mode))
        if partial:
            monkeypatch.delenv("CEDR_HIP_EXCHANGE")
        out = np.zeros_like(v.Qm)

        def body(r, comm):
            tree = ca.Tree.from_arrays(root, kid0, kid1, rank, cell)
            if partial:
                tree = tree.prune(r)
            q = ca.QLT(comm.parallel(ca, r), nleaf, tree)
            for t in types:
                q.declare_tracer(t, 0)
            q.end_tracer_declarations()
            q.finish_setup()
            if mode != 0:  # (0: the planner's choice -- level mode only if a pinned node sits inside a single-rank subtree)
                assert q.plan()["level_mode"] == (mode == 2)
            g = q.get_owned_glblcells()
            d = [ca.DeviceArray.from_numpy(a) for a in
                 (v.rhom[g], v.Qm[:, g], v.Qm_min[:, g], v.Qm_max[:, g], v.Qm_prev[:, g])]
            q.set_rhom(d[0])
            for _ in range(2):
                q.set_Qm(d[1], d[2], d[3], d[4])
                q.run()
            od = ca.DeviceArray(len(types) * len(g))
            q.get_Qm(od)
            ca.synchronize()
            out[:, g] = od.numpy((len(types), len(g)))
        T.run_ranks(nranks, body)
        assert np.array_equal(out, want), (mode, partial)


@pytest.mark.parametrize("kind,nleaf,nranks,how", [("merge", 500, 4, "scattered"), ("skew", 300, 3, "blocks"),
                                                   ("chain", 64, 5, "scattered"), ("merge", 4000, 8, "blocks")])
@pytest.mark.parametrize("transpose", [False, True])
def test_bfb_random_caller_trees_ranks_as_threads(gpu, oracle, kind, nleaf, nranks, how, transpose, monkeypatch):
    """BfbTreeAllReducer over irregular trees and scattered ownership, all three exchange forms:
    every rank gets the bits of the single-rank sum over the same tree."""
    import random_trees as RT
    ca = gpu
    nfield = 3
    rng = np.random.default_rng(nleaf + nranks)
    root, kid0, kid1, rank, cell = RT.random_tree(rng, nleaf, kind)
    rank = RT.assign_ranks(rng, nleaf, rank, nranks, how)
    glob = rng.random((nleaf, nfield)) - 0.5  # [global cell id][field]
    lay = lambda a: np.ascontiguousarray(a.T if transpose else a).ravel()
    r1 = np.full(len(kid0), -1, np.int32)
    r1[:nleaf] = 0
    leaves1 = _bfb_local_leaves(ca.Tree.from_arrays(root, kid0, kid1, r1, cell), nleaf, 0)
    want = oracle.bfb_allreduce(oracle.Tree.from_arrays(root, kid0, kid1, r1, cell), nleaf, 1, nfield, transpose,
                                [lay(glob[leaves1])])[0]
    for mode, partial in ((1, False), (2, False), (2, True)):
        monkeypatch.setenv("CEDR_HIP_EXCHANGE", str(mode))
        if partial:
            monkeypatch.delenv("CEDR_HIP_EXCHANGE")
        got = [None] * nranks

        def body(r, comm):
            tree = ca.Tree.from_arrays(root, kid0, kid1, rank, cell)
            if partial:
                tree = tree.prune(r)
            mine = _bfb_local_leaves(tree, nleaf, r)
            b = ca.BfbTreeAllReducer(comm.parallel(ca, r), tree, nleaf, nfield)
            assert b.nlocal() == len(mine)
            sd = ca.DeviceArray.from_numpy(lay(glob[mine]))
            rd = ca.DeviceArray(nfield)
            for _ in range(2):
                b.allreduce(sd, rd, transpose)
            ca.synchronize()
            got[r] = rd.numpy()
        T.run_ranks(nranks, body)
        for r in range(nranks):
            assert np.array_equal(got[r], want), (mode, partial, r)
