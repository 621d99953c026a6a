"""Multi-rank data path on one GPU: P ranks as P threads (tests/threaded_ranks.py),
checked bit-for-bit against the single-rank oracle -- QLT results may not depend on
the rank count or on who owns which cell (reference cedr_qlt.cpp:745-772 runs the same
harness on any communicator size)."""
import numpy as np
import pytest

import common
import randomized as R
import threaded_ranks as T

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
    assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()
    assert R.check(trs, v, got, False, verbose=False)[0] == 0


def test_qlt_bound_arrays_ranks_as_threads(gpu, oracle):
    """Zero-copy mode on 4 ranks: every rank binds its own slices."""
    ca = gpu
    nranks, ncells = 4, 3000
    trs = R.tracers_vector()
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, ncells, seed=5)
    want = common.run_oracle_qlt(oracle, (False, False), ncells, 1, types, v)
    out = np.zeros_like(v.Qm)

    def body(rank, comm):
        q = ca.QLT(comm.parallel(ca, rank), ncells, ca.Tree.make_1d(ncells, nranks, False, False))
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
