"""GPU parity of CAAS against the CPU oracle. The oracle sums cells serially
(Kokkos-Serial order); the HIP path sums in a fixed tree order, so sums agree to
rounding and the reconstructed masses to 1e-13 relative (north star tolerance)."""
import numpy as np
import pytest

import common
import randomized as R

pytestmark = pytest.mark.gpu
TOL = 1e-13


def _rel(a, b):
    """Largest per-tracer relative difference: max_i |a - b| / max_i |b| of each tracer (row)."""
    a, b = np.atleast_2d(a), np.atleast_2d(b)
    return (np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1e-300)).max()


# caas::test::unittest (cedr_caas.cpp:381-393): nlclcells in {1,2,4,11}, own reducer or
# not, external memory or not; plus sizes that span several chunks, odd and even.
@pytest.mark.parametrize("ncells", [1, 2, 4, 11, 1000, 4097, 20000])
@pytest.mark.parametrize("external_memory", [False, True])
def test_caas_harness(gpu, oracle, ncells, external_memory):
    ca = gpu
    trs = R.caas_tracers(R.tracers_vector())
    v = R.make_values(trs, ncells, seed=300 + ncells)
    types = [t.problem_type for t in trs]
    want, _ = common.run_oracle_caas(oracle, [ncells], types, v)
    got = common.run_hip_caas(ca, ncells, types, v, external_memory)
    assert _rel(got, want) <= TOL
    nerr, msgs = R.check(trs, v, got)
    assert nerr == 0, msgs[:5]
    # perturbation 0 must come back exactly
    for t in trs:
        if t.no_change_should_hold:
            assert np.array_equal(got[t.idx], v.Qm_prev[t.idx])


@pytest.mark.parametrize("ncells,accum", [(12, 3), (12, 2), (11, 1), (3000, 3)])
def test_caas_user_reducer(gpu, oracle, ncells, accum):
    """UserAllReducer path (cedr_caas.cpp:137-168,263-266) with the reference test's
    reducer (cedr_caas.cpp:304-321) run on the host through device copies."""
    ca = gpu
    trs = R.caas_tracers(R.tracers_vector())
    v = R.make_values(trs, ncells, seed=77)
    types = [t.problem_type for t in trs]
    L = ca.lib()

    def reducer(send, recv, nlocal, nfld):
        s = np.empty(nlocal * nfld)
        L.cedr_memcpy_d2h(s.ctypes.data, send, s.nbytes)
        r = np.empty(nfld)
        for k in range(nfld):
            acc = s[nlocal * k]
            for i in range(1, nlocal):
                acc += s[nlocal * k + i]
            r[k] = acc
        L.cedr_memcpy_h2d(recv, r.ctypes.data, r.nbytes)
        return 0

    want, _ = common.run_oracle_caas(oracle, [ncells], types, v, accum=accum)
    got = common.run_hip_caas(ca, ncells, types, v, user_reducer=reducer, accum=accum)
    assert np.array_equal(got, want)  # same summation order as the oracle => bit for bit


# The production pairing: CAAS + the library's BFB tree reducer. (groups, cells per group): tiles of 183-192 leaves;
# accum 4 = the profile's case; 2, 4, 8, 16, 32: group sums in registers along the lanes (park: the LDS form instead,
# where a tile's cells fit); 3 and 1: odd groups (8-byte loads, LDS form); 6, 10: LDS form, 10 with more than 1024 cells
# per tile (loop form); 44, 64: too many cells for a tile's LDS -> not fused; 1500 groups: one stage of micro tiles
# under the top tile; 40 groups: a single tile.
_BFB_CASES = [(5000, 4, True), (5000, 3, True), (3000, 1, True), (4097, 2, True), (2500, 10, True), (40000, 4, True),
              (2304, 44, False), (1500, 4, True), (40, 4, None), (12345, 6, True), (3000, 16, True), (2001, 8, True),
              (1500, 32, True), (700, 64, False)]


@pytest.mark.parametrize("ngroups,accum,expect_fused,park",
                         [c + (False,) for c in _BFB_CASES] +
                         # park: the LDS form of the fused kernel also for the group sizes that are otherwise summed in
                         # registers (with 16 and more cells per group a tile's cells rarely fit the LDS: fused or not)
                         [(g, a, e if a < 16 else None, True) for g, a, e in _BFB_CASES if a in (2, 4, 8, 16, 32)])
@pytest.mark.parametrize("imbalanced", [False, True])
def test_caas_bfb_pairing_fused_bitwise(gpu, oracle, monkeypatch, ngroups, accum, expect_fused, imbalanced, park):
    """CAAS whose UserAllReducer is the BFB tree reducer (cedr_caas.cpp:137-168,263-266 +
    cedr_bfb_tree_allreduce.cpp:79-159): the fused kernel (typed pairing) gives the bits of the two separate
    launches (callback pairing) and of the restatement with the oracle's tree sum."""
    ca = gpu
    if not common.default_knobs():
        expect_fused = None  # (env matrix: another code path may be forced; the bits must hold anyway)
    if park:
        monkeypatch.setenv("CEDR_HIP_CAAS_BFB_P2", "0")
    trs = R.caas_tracers(R.tracers_vector())
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, ngroups * accum, seed=ngroups + accum)
    want = common.caas_bfb_expected(oracle, v, types, ngroups, accum, imbalanced)
    cb, f0 = common.run_hip_caas_bfb(ca, ngroups, accum, types, v, typed=False, imbalanced=imbalanced)
    assert not f0
    assert np.array_equal(cb, want)
    got, fused = common.run_hip_caas_bfb(ca, ngroups, accum, types, v, typed=True, imbalanced=imbalanced)
    assert expect_fused is None or fused == expect_fused
    assert np.array_equal(got, want)
    nerr, msgs = R.check(trs, v, got)
    assert nerr == 0, msgs[:5]


def test_caas_bfb_pairing_fuse_knob_and_reruns(gpu, oracle, monkeypatch):
    """CEDR_HIP_CAAS_BFB_FUSE=0 keeps the two launches; repeated run() calls on re-packed inputs and the
    set_Qm_source form give the same bits."""
    ca = gpu
    trs = R.caas_tracers(R.tracers_vector())
    types = [t.problem_type for t in trs]
    ngroups, accum = 6000, 4
    v = R.make_values(trs, ngroups * accum, seed=11)
    want = common.caas_bfb_expected(oracle, v, types, ngroups, accum)
    dflt = common.default_knobs()
    for kw in (dict(nrepeat=3), dict(qm_source=True)):
        got, fused = common.run_hip_caas_bfb(ca, ngroups, accum, types, v, typed=True, **kw)
        assert (fused or not dflt) and np.array_equal(got, want)
    monkeypatch.setenv("CEDR_HIP_CAAS_BFB_FUSE", "0")
    got, fused = common.run_hip_caas_bfb(ca, ngroups, accum, types, v, typed=True)
    assert not fused and np.array_equal(got, want)


def test_caas_qm_source_matches_in_place(gpu):
    ca = gpu
    trs = [R.Tracer(i, 7, 1) for i in range(5)]
    v = R.make_values(trs, 8192 + 6, seed=3)
    a = common.run_hip_caas(ca, v.ncells, [7] * 5, v)
    b = common.run_hip_caas(ca, v.ncells, [7] * 5, v, qm_source=True)
    assert np.array_equal(a, b)


def test_caas_errors(gpu):
    ca = gpu
    with pytest.raises(ca.CedrError):
        ca.CAAS(ca.Parallel.self_(), 0)  # cedr_caas.cpp:46
    c = ca.CAAS(ca.Parallel.self_(), 4)
    with pytest.raises(ca.CedrError):
        c.declare_tracer(4, 0)  # !shapepreserve, cedr_caas.cpp:52-53
    with pytest.raises(ca.CedrError):
        c.declare_tracer(7, 1)  # rhomidx > 0
    with pytest.raises(ca.CedrError):
        c.end_tracer_declarations()  # 0 tracers, cedr_caas.cpp:62


@pytest.mark.parametrize("ncells", [1, 2, 777, 4096, 10001])
def test_caas_bound_arrays_in_place(gpu, oracle, ncells):
    """Zero-copy mode: CAAS limits the caller's own [tracer][cell] arrays
    in place and gives bit-for-bit what the packed path gives (same kernels, same order)."""
    ca = gpu
    trs = R.caas_tracers(R.tracers_vector())
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, ncells, seed=31 + ncells)
    packed = common.run_hip_caas(ca, ncells, types, v)
    c = ca.CAAS(ca.Parallel.self_(), ncells)
    for t in types:
        c.declare_tracer(t, 0)
    c.end_tracer_declarations()
    c.finish_setup()
    d = [ca.DeviceArray.from_numpy(a) for a in (v.Qm, v.Qm_min, v.Qm_max, v.Qm_prev)]
    c.bind_arrays(d[0], d[1], d[2], d[3])
    c.run()
    ca.synchronize()
    got = d[0].numpy(v.Qm.shape)
    assert np.array_equal(got, packed)
    assert np.array_equal(d[1].numpy(v.Qm.shape), v.Qm_min)  # inputs untouched
    want, _ = common.run_oracle_caas(oracle, [ncells], types, v)
    assert _rel(got, want) <= TOL
    with pytest.raises(ca.CedrError):
        c.bind_arrays(d[0], d[1], d[2], None)  # conserving tracers need Qm_prev
