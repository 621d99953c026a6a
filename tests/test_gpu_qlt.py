"""GPU parity of QLT: the HIP path (through the C ABI) against the CPU oracle on
the reference's own randomized problems (cedr_test_randomized.cpp), bit for bit,
plus the reference's property checks with its tolerances."""
import os

import numpy as np
import pytest

import common
import random_trees
import randomized as R

pytestmark = pytest.mark.gpu

CST = 7


def _types(trs):
    return [t.problem_type for t in trs]


# unittest_QLT (cedr_qlt.cpp:745-772) uses sizes {1,2,7,21} x nranks; here single
# rank sizes plus larger ones that exercise several stages of tiles.
@pytest.mark.parametrize("ncells", [1, 2, 3, 7, 21, 111, 1000, 1024, 5000])
@pytest.mark.parametrize("imbalanced", [False, True])
@pytest.mark.parametrize("prefer", [False, True])
def test_qlt_harness_bitwise(gpu, oracle, ncells, imbalanced, prefer):
    ca = gpu
    trs = R.tracers_vector()
    v = R.make_values(trs, ncells, seed=1000 + ncells)
    want = common.run_oracle_qlt(oracle, (False, imbalanced), ncells, 1, _types(trs), v, prefer)
    tree = ca.Tree.make_1d(ncells, 1, False, imbalanced)
    q = common.make_hip_qlt(ca, tree, ncells, _types(trs), prefer, external_memory=imbalanced)
    got = common.run_hip_qlt(ca, q, v, nrepeat=2)
    # integer tree indexing: local cell order identical
    assert np.array_equal(q.get_owned_glblcells(),
                          oracle.OracleQLT(oracle.Tree.make_1d(ncells, 1, False, imbalanced), ncells)
                          .owned_glblcells(0))
    bad = np.argwhere(got != want)
    assert bad.size == 0, "first mismatch (tracer, cell) %s: %r vs %r" % (
        bad[0], got[tuple(bad[0])], want[tuple(bad[0])])
    if ncells <= 1024:  # the reference itself fails 'c nn' mass at >= 2k cells
        nerr, msgs = R.check(trs, v, got, prefer)
        assert nerr == 0, msgs[:5]


def test_qlt_config_A(gpu, oracle):
    """BASELINE config A: 1024 cells x 4 cst tracers; buffer sizes as the reference's."""
    ca = gpu
    trs = [R.Tracer(i, CST, 1) for i in range(4)]
    v = R.make_values(trs, 1024, seed=7)
    tree = ca.Tree.make_1d(1024)
    q = common.make_hip_qlt(ca, tree, 1024, [CST] * 4)
    assert q.get_buffers_sizes() == (2047 * 17, 2047 * 4)  # 34799, 8188
    got = common.run_hip_qlt(ca, q, v)
    want = common.run_oracle_qlt(oracle, (False, False), 1024, 1, [CST] * 4, v)
    assert np.array_equal(got, want)
    rel = np.abs(got.sum(1) - v.Qm_prev.sum(1)) / np.abs(v.Qm_prev).sum(1)
    assert rel.max() < 1e-13
    assert (got >= v.Qm_min).all() and (got <= v.Qm_max).all()


@pytest.mark.parametrize("micro", ["0", "1", "cut0", "cut1", "cut5"])
@pytest.mark.parametrize("tile", [2, 3, 16, 128, 256])
def test_qlt_tile_size_invariance(gpu, oracle, tile, micro, monkeypatch):
    """Results cannot depend on how the tree is cut into tiles (micro tiles or not)."""
    ca = gpu
    monkeypatch.setenv("CEDR_HIP_TILE_LEAVES", str(tile))
    monkeypatch.setenv("CEDR_HIP_MICRO", "0" if micro == "0" else "1")
    if micro.startswith("cut"):
        monkeypatch.setenv("CEDR_HIP_MICRO_CUT", micro[3:])
    trs = R.tracers_vector()
    v = R.make_values(trs, 333, seed=5)
    want = common.run_oracle_qlt(oracle, (False, True), 333, 1, _types(trs), v)
    q = common.make_hip_qlt(ca, ca.Tree.make_1d(333, 1, False, True), 333, _types(trs))
    got = common.run_hip_qlt(ca, q, v)
    assert np.array_equal(got, want)
    # a balanced tree too (micro shapes (a,b) and (a,(b,c)) dominate there)
    v2 = R.make_values(trs, 1500, seed=6)
    want2 = common.run_oracle_qlt(oracle, (False, False), 1500, 1, _types(trs), v2)
    q2 = common.make_hip_qlt(ca, ca.Tree.make_1d(1500), 1500, _types(trs))
    assert np.array_equal(common.run_hip_qlt(ca, q2, v2), want2)


def test_qlt_interleaved_types_and_errors(gpu, oracle):
    ca = gpu
    types = [4, 7, 8, 2, 9, 5, 3, 6, 7, 8]  # bidx2trcr is a non-trivial permutation
    trs = [R.Tracer(i, oracle.lib().or_canonical_problem_type(t), 1) for i, t in enumerate(types)]
    v = R.make_values(trs, 200, seed=11)
    tree = ca.Tree.make_1d(200)
    q = common.make_hip_qlt(ca, tree, 200, types)
    oq = oracle.OracleQLT(oracle.Tree.make_1d(200), 200)
    oq.declare(types)
    md, omd = q.metadata(), oq.meta()
    for k in omd:
        assert np.array_equal(md[k], omd[k]), k
    for i, t in enumerate(types):
        assert q.get_problem_type(i) == oracle.lib().or_canonical_problem_type(t)
    got = common.run_hip_qlt(ca, q, v)
    want = common.run_oracle_qlt(oracle, (False, False), 200, 1, types, v)
    assert np.array_equal(got, want)
    with pytest.raises(ca.CedrError):
        q.declare_tracer(7, 0)  # after end_tracer_declarations (cedr_qlt.cpp:278-279)
    q2 = ca.QLT(ca.Parallel.self_(), 200, tree)
    with pytest.raises(ca.CedrError):
        q2.declare_tracer(7, 1)  # rhomidx > 0 (cedr_qlt.cpp:280)
    with pytest.raises(ca.CedrError):
        q2.declare_tracer(1, 0)  # conserve alone is invalid (cedr_qlt.cpp:94)


# (sizes chosen so that the last micro-subtree of the last tile has every shape: the
# kernels may not read past the end of a caller row, which has no padding)
@pytest.mark.parametrize("ncells", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 64, 65, 111, 193, 1000, 5000])
@pytest.mark.parametrize("in_place", [False, True])
def test_qlt_bound_arrays_zero_copy(gpu, oracle, ncells, in_place):
    """Zero-copy mode: run() sweeps the caller's own [tracer][cell] arrays
    and writes the masses to Qm_out (or over Qm): bit-for-bit the packed path's result, for
    every problem type (the t-only tracers are the ones still packed inside run())."""
    ca = gpu
    trs = R.tracers_vector()
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, ncells, seed=777 + ncells)
    want = common.run_oracle_qlt(oracle, (False, False), ncells, 1, types, v)
    q = common.make_hip_qlt(ca, ca.Tree.make_1d(ncells), ncells, types)
    d = [ca.DeviceArray.from_numpy(a) for a in (v.rhom, v.Qm, v.Qm_min, v.Qm_max, v.Qm_prev)]
    out = d[1] if in_place else ca.DeviceArray(v.Qm.size)
    q.bind_arrays(d[0], d[1], d[2], d[3], d[4], out)
    q.run()
    ca.synchronize()
    assert np.array_equal(out.numpy(v.Qm.shape), want)
    if not in_place:
        assert np.array_equal(d[1].numpy(v.Qm.shape), v.Qm)  # inputs untouched
        q.run()  # a second run on the same inputs repeats the first
        ca.synchronize()
        assert np.array_equal(out.numpy(v.Qm.shape), want)
    # back to the packed path on the same object
    q.bind_arrays(None, None, None, None, None, None)
    got = common.run_hip_qlt(ca, q, v)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("kind,nleaf", [("merge", 2), ("merge", 17), ("merge", 500), ("merge", 3001),
                                        ("chain", 3), ("chain", 200), ("chain", 1500),
                                        ("skew", 64), ("skew", 2000)])
def test_qlt_caller_trees_bitwise(gpu, oracle, kind, nleaf):
    """Trees a caller might hand over (cedr_tree_caller.hpp:12-30) rather than the 1-D mesh
    bisection: irregular, very deep, skewed; global cell ids in random order."""
    ca = gpu
    rng = np.random.default_rng(1000 + nleaf)
    root, kid0, kid1, rank, cell = random_trees.random_tree(rng, nleaf, kind)
    trs = R.tracers_vector()
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, nleaf, seed=nleaf)
    ot = oracle.Tree.from_arrays(root, kid0, kid1, rank, cell)
    oq = oracle.OracleQLT(ot, nleaf, 1, False)
    oq.declare(types)
    g = oq.owned_glblcells(0)
    oq.set(0, v.rhom[g], v.Qm[:, g], v.Qm_min[:, g], v.Qm_max[:, g], v.Qm_prev[:, g])
    oq.run()
    want = np.zeros_like(v.Qm)
    want[:, g] = oq.get(0)
    q = common.make_hip_qlt(ca, ca.Tree.from_arrays(root, kid0, kid1, rank, cell), nleaf, types)
    assert np.array_equal(q.get_owned_glblcells(), g)
    got = common.run_hip_qlt(ca, q, v, nrepeat=2)
    assert np.array_equal(got, want)
    # (the harness's property tolerances are for trees of logarithmic depth; a 1500-level
    # caterpillar accumulates more roundoff in the reference algorithm itself)
    if kind == "merge":
        assert R.check(trs, v, got, False, verbose=False)[0] == 0


def test_array_arguments_are_validated(gpu):
    """A wrong element type, a short array or a missing Qm_prev is an error message, never a kernel
    reading the wrong bytes (the inline DeviceOp traps on a missing Qm_prev: cedr_qlt_inl.hpp:53-54)."""
    ca = gpu
    n, types = 100, [7, 2]
    q = common.make_hip_qlt(ca, ca.Tree.make_1d(n), n, types)
    d = ca.DeviceArray(n * len(types))
    with pytest.raises(ca.CedrError):
        q.set_Qm(d, d, d, None)                      # tracer 0 conserves
    with pytest.raises(ca.CedrError):
        q.set_Qm(d, d, ca.DeviceArray(n), d)         # too short
    with pytest.raises(ca.CedrError):
        q.set_rhom(ca.DeviceArray(n, np.int32))      # not doubles
    try:
        import torch
    except ImportError:
        return
    dev = torch.device("cuda", 0)
    with pytest.raises(ca.CedrError):
        q.set_rhom(torch.zeros(n, dtype=torch.float32, device=dev))
    with pytest.raises(ca.CedrError):
        q.set_rhom(torch.zeros(n, dtype=torch.float64))  # host tensor
    with pytest.raises(ca.CedrError):
        q.set_rhom(torch.zeros(2 * n, dtype=torch.float64, device=dev)[::2])  # strided


def test_tight_buffers(gpu, oracle):
    """set_buffers with the TIGHT sizes (what this build touches, about half of the reference's
    figure) gives the same bits as library-owned memory."""
    ca = gpu
    n = 3000
    trs = R.tracers_vector()[:12]
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, n, seed=11)
    q = ca.QLT(ca.Parallel.self_(), n, ca.Tree.make_1d(n))
    for t in types:
        q.declare_tracer(t, 0)
    q.end_tracer_declarations()
    b1, b2 = q.get_buffers_sizes()
    t1, t2 = q.get_buffers_sizes_tight()
    assert t1 < 0.6 * b1 and t2 < 0.6 * b2
    q.set_buffers(ca.DeviceArray(t1), ca.DeviceArray(t2))
    q.finish_setup()
    got = common.run_hip_qlt(ca, q, v)
    want = common.run_oracle_qlt(oracle, (False, False), n, 1, types, v)
    assert np.array_equal(got, want)


def test_sweep_down_event_orders_a_second_stream(gpu, oracle):
    """QLT::sweep_down_event: CAAS on a second stream starts when QLT::run reaches its first-stage
    root->leaves sweep. Both results must be what the two runs give one after the other."""
    import torch
    ca = gpu
    nc = 20000
    trs = R.tracers_vector()[:12]
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, nc, seed=31)
    want = common.run_oracle_qlt(oracle, (False, False), nc, 1, types, v)
    q = common.make_hip_qlt(ca, ca.Tree.make_1d(nc, 1), nc, types)
    ctr = R.caas_tracers(trs)
    ctypes_ = [t.problem_type for t in ctr]
    cv = R.make_values(ctr, nc, seed=32)
    cwant, _ = common.run_oracle_caas(oracle, [nc], ctypes_, cv)
    c = ca.CAAS(ca.Parallel.self_(), nc)
    for t in ctypes_:
        c.declare_tracer(t, 0)
    c.end_tracer_declarations()
    c.finish_setup()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    q.set_stream(sa.cuda_stream)
    c.set_stream(sb.cuda_stream)
    ev = q.sweep_down_event()
    assert ev and q.sweep_down_event() == ev
    d = [ca.DeviceArray.from_numpy(a) for a in (v.rhom, v.Qm, v.Qm_min, v.Qm_max, v.Qm_prev)]
    e = [ca.DeviceArray.from_numpy(a) for a in (cv.rhom, cv.Qm, cv.Qm_min, cv.Qm_max, cv.Qm_prev)]
    q.set_rhom(d[0])
    c.set_rhom(e[0])
    for _ in range(3):
        q.set_Qm(d[1], d[2], d[3], d[4])
        c.set_Qm(e[1], e[2], e[3], e[4])
        q.run()
        ca.stream_wait_event(sb.cuda_stream, ev)
        c.run()
    qo, co = ca.DeviceArray(v.Qm.size), ca.DeviceArray(cv.Qm.size)
    q.get_Qm(qo)
    c.get_Qm(co)
    ca.synchronize()
    assert np.array_equal(qo.numpy(v.Qm.shape), want)
    assert common.rel_per_tracer(co.numpy(cv.Qm.shape), cwant).max() <= 1e-13


def test_qlt_on_a_cu_masked_stream(gpu, oracle):
    """cedr_stream_create_cu_mask: a stream restricted to half of the compute units (every other group of eight).
    QLT::run on it gives the same bits; an empty argument list is refused."""
    ca = gpu
    trs = R.tracers_vector()
    types = [t.problem_type for t in trs]
    ncells = 3000
    v = R.make_values(trs, ncells, seed=77)
    want = common.run_oracle_qlt(oracle, (False, False), ncells, 1, types, v)
    q = common.make_hip_qlt(ca, ca.Tree.make_1d(ncells), ncells, types)
    ms = ca.MaskedStream([0x00FF00FF] * 8)
    q.set_stream(ms.handle)
    got = common.run_hip_qlt(ca, q, v)
    assert np.array_equal(got, want)
    q.set_stream(0)
    ms.close()
    with pytest.raises(ca.CedrError):
        ca.MaskedStream([])


@pytest.mark.parametrize("limit,max_top_inputs", [(None, 2048), ("65536", 512), ("40000", 256), ("20000", 0)])
def test_top_tile_follows_the_lds_the_device_grants(gpu, oracle, limit, max_top_inputs, monkeypatch):
    """The fused top tile keeps 72 B of LDS per internal node (+ ~10 KB for the apex and tables): 2048 inputs need
    gfx950's 160 KB. The planner sizes it from what the device grants a workgroup (CEDR_HIP_LDS_LIMIT plans as for a
    smaller device): a smaller top tile, or none below 24 KB (one launch per stage) -- never a launch that asks for
    more LDS than there is. Same bits either way."""
    ca = gpu
    if limit:
        monkeypatch.setenv("CEDR_HIP_LDS_LIMIT", limit)
    ncells = 40000  # 219 first-stage tiles -> 3504 outputs: more than any top tile takes, so a second stage exists
    trs = R.tracers_vector()[:12]
    v = R.make_values(trs, ncells, seed=3)
    want = common.run_oracle_qlt(oracle, (False, False), ncells, 1, _types(trs), v)
    q = common.make_hip_qlt(ca, ca.Tree.make_1d(ncells), ncells, _types(trs))
    tops = [st for st in q.plan()["local"] if st["top"]]
    if "CEDR_HIP_TOP" in os.environ or "CEDR_HIP_TOP_MAX" in os.environ:
        pass  # (an env-matrix run that sets the top tile's knobs itself: only the bits are checked)
    elif max_top_inputs == 0:
        assert not tops
    else:
        assert len(tops) == 1
        nin = int(tops[0]["tiles"][0][1])  # the tile's input count
        assert 2 <= nin <= max_top_inputs, nin
        if limit:
            assert 72 * (nin - 1) + 10 * 1024 <= int(limit)
    assert np.array_equal(common.run_hip_qlt(ca, q, v), want)


def test_device_lds_does_not_cap_the_top_tile(gpu, monkeypatch):
    """engine::plan_options bounds the fused top tile by the LDS the device grants a workgroup. An MI355X grants
    160 KB: the plan a QLT runs (made with the device in view) must be the plan made on the host alone
    (cedr_plan_create never touches a GPU, so nothing caps it) -- at a size whose top tile takes the full 2048 inputs.
    A runtime that reported less would shrink the tile silently otherwise (the library now also says so on stderr)."""
    ca = gpu
    for k in ("CEDR_HIP_TOP_MAX", "CEDR_HIP_LDS_LIMIT", "CEDR_HIP_TOP", "CEDR_HIP_MICRO", "CEDR_HIP_MICRO_CUT",
              "CEDR_HIP_MICRO_CUT2", "CEDR_HIP_TILE_LEAVES", "CEDR_HIP_EXCHANGE"):
        monkeypatch.delenv(k, raising=False)
    n = 1500000
    tree = ca.Tree.make_1d(n)
    q = ca.QLT(ca.Parallel.self_(), n, tree)
    q.declare_tracer(7, 0)
    q.end_tracer_declarations()
    dev_plan = q.plan()
    hp = ca.lib().cedr_plan_create(tree.h, n, 0, 1, 0)
    host_plan = ca.dump_plan(hp, 1)
    ca.lib().cedr_plan_free(hp)
    shape = lambda p: [(len(st["tiles"]), bool(st["top"]), int(st["tiles"][0][1])) for st in p["local"]]
    assert shape(dev_plan) == shape(host_plan)
    assert dev_plan["local"][-1]["top"] and int(dev_plan["local"][-1]["tiles"][0][1]) == 2048
