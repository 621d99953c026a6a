"""Integration: the reference's 1-D semi-Lagrangian transport demo
(cedr_test_1d_transport.cpp:264-330, run serially with -nc 111 by ctest
cedr-test-t1d) with QLT-nonnegative, QLT and CAAS as the limiter, HIP library vs
oracle over 3.17*ncells time steps. Each step's output feeds the next step, so the
QLT fields must stay bit-identical throughout."""
import numpy as np
import pytest

import transport1d as T

pytestmark = pytest.mark.gpu
NC = 111


def _hip_stepper(ca, kind, ptype, area):
    p = ca.Parallel.self_()
    cdr = ca.QLT(p, NC, ca.Tree.make_1d(NC)) if kind == "qlt" else ca.CAAS(p, NC)
    cdr.declare_tracer(ptype, 0)
    cdr.end_tracer_declarations()
    cdr.finish_setup()
    cdr.set_rhom(ca.DeviceArray.from_numpy(area))
    d = [ca.DeviceArray(NC) for _ in range(5)]

    def step(Qm, lo, hi, prev):
        for buf, a in zip(d, (Qm, lo, hi, prev)):
            buf.upload(a)
        cdr.set_Qm(d[0], d[1], d[2], d[3])
        cdr.run()
        cdr.get_Qm(d[4])
        ca.synchronize()
        return d[4].numpy()
    return step


def _oracle_stepper(o, kind, ptype, area):
    if kind == "qlt":
        cdr = o.OracleQLT(o.Tree.make_1d(NC), NC)
        cdr.declare([ptype])
    else:
        cdr = o.OracleCAAS([NC], [ptype])
    first = [True]

    def step(Qm, lo, hi, prev):
        cdr.set(0, area if first[0] else None, Qm[None], lo[None], hi[None], prev[None])
        first[0] = False
        cdr.run()
        return cdr.get(0)[0]
    return step


@pytest.mark.parametrize("kind,ptype", [("qlt", 1 | 8), ("qlt", 1 | 2), ("caas", 1 | 2)])
def test_transport1d(gpu, oracle, kind, ptype):
    prob = T.Problem1D(NC)
    y0 = np.concatenate([T.initial_condition(prob.xcp[:NC]), [0.0]])
    y0[NC] = y0[0]
    nsteps = int(3.17 * NC)
    want = prob.cycle(nsteps, y0, _oracle_stepper(oracle, kind, ptype, prob.area))
    got = prob.cycle(nsteps, y0, _hip_stepper(gpu, kind, ptype, prob.area))
    if kind == "qlt":
        assert np.array_equal(got, want)
    else:
        assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()
    m0, m1 = (y0[:NC] * prob.area).sum(), (got[:NC] * prob.area).sum()
    assert abs(m1 - m0) <= 1e-12 * abs(m0)  # mass is conserved through all the steps
    if ptype & 2:
        assert got.min() >= y0.min() - 1e-12 and got.max() <= y0.max() + 1e-12  # no new extrema
    else:
        assert got.min() >= -1e-14  # nonnegative
