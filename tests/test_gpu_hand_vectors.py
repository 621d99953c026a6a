"""The hand-derived vectors (tests/golden/hand_derived_vectors.md) on the DEVICE: the public
cedr::local 2-D solver, and the node problem exactly as the sweep kernels evaluate it
(cedr_node_math.hpp: a = (1,1) specialisation, mask-logic edge choice, lockstep pair form).
Plus a randomised bitwise comparison of that device code with the oracle's solve_node_problem."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
V = json.load(open(os.path.join(HERE, "golden", "hand_derived_vectors.json")))


def test_qp_2d_on_device(gpu):
    ca = gpu
    for clip, early in ((True, True), (True, False), (False, False)):
        cs = [c for c in V["solve_1eq_bc_qp_2d"] if c["clip"] == clip and c["early"] == early]
        if not cs:
            continue
        g = lambda k: np.array([c[k] for c in cs], np.float64)
        info, x = ca.local_batch(ca.LOCAL_BC_QP_2D, g("w"), np.ones((len(cs), 2)), g("b"), g("xlo"), g("xhi"), g("y"),
                                 clip=clip, early_exit_on_tol=early)
        for i, c in enumerate(cs):
            assert info[i] == c["info"] and list(x[i]) == c["x"], c["name"]


def test_node_problem_on_device(gpu):
    ca = gpu
    for c in V["solve_node_problem"]:
        x = ca.node_problem_batch(c["pt"], [c["rhom"]], [c["pd"]], [c["Qm"]], [c["rhom0"]], [c["k0d"]],
                                  [c["rhom1"]], [c["k1d"]])
        assert list(x[0]) == c["x"], c["name"]
    sp = [c for c in V["solve_node_problem"] if c["pt"] & 2]
    sp = sp[:len(sp) // 2 * 2]
    g = lambda k: [c[k] for c in sp]
    x = ca.node_problem_batch(7, g("rhom"), g("pd"), g("Qm"), g("rhom0"), g("k0d"), g("rhom1"), g("k1d"), pair=True)
    for i, c in enumerate(sp):
        assert list(x[i]) == c["x"], "lockstep pair: " + c["name"]


def random_problems(n, seed):
    """Scenarios: quick exit, interior, edges, infeasible either way, exact corners, y outside."""
    rng = np.random.default_rng(seed)
    rh0, rh1 = 0.5 * (1 + rng.random(n)), 0.5 * (1 + rng.random(n))
    k = np.zeros((2, n, 3))
    for j in range(2):
        lo = -0.75 + rng.random(n)
        hi = lo + rng.random(n)
        k[j, :, 0], k[j, :, 2] = lo, hi
        k[j, :, 1] = lo + (hi - lo) * rng.random(n)
    sc = np.arange(n) % 8
    k[0, sc == 5, 1] = k[0, sc == 5, 0] - rng.random((sc == 5).sum())  # a kid outside its bounds
    pd = k[0] + k[1]
    u = rng.random(n)
    Qm = np.select([sc == 0, sc == 1, sc == 2, sc == 3, sc == 4, sc == 6],
                   [pd[:, 1], pd[:, 0] - 0.3 * u, pd[:, 2] + 0.3 * u, pd[:, 0], pd[:, 2], pd[:, 1] + 1e-3 * (u - 0.5)],
                   pd[:, 0] + (pd[:, 2] - pd[:, 0]) * u)
    return rh0 + rh1, pd, Qm, rh0, k[0], rh1, k[1]


@pytest.mark.parametrize("prefer", [False, True])
def test_device_node_math_matches_oracle_bitwise(gpu, oracle, prefer):
    ca = gpu
    n = 4000
    rhom, pd, Qm, rh0, k0, rh1, k1 = random_problems(n, seed=31)
    want = np.array([oracle.solve_node_problem(7, rhom[i], pd[i], Qm[i], rh0[i], k0[i], rh1[i], k1[i], prefer)
                     for i in range(n)])
    for pair in (False, True):
        got = ca.node_problem_batch(7, rhom, pd, Qm, rh0, k0, rh1, k1, prefer=prefer, pair=pair)
        assert np.array_equal(got.view(np.int64), want.view(np.int64)), "pair=%s" % pair
    # consistent-only (t) and nonnegative types through the same dispatch
    q = np.column_stack([pd[:, 0] / rhom, pd[:, 1], pd[:, 2] / rhom])
    q0 = np.column_stack([k0[:, 0] / rh0, k0[:, 1], k0[:, 2] / rh0])
    q1 = np.column_stack([k1[:, 0] / rh1, k1[:, 1], k1[:, 2] / rh1])
    for pt, a, b, c in ((4, q, q0, q1), (8, np.abs(pd), np.abs(k0), np.abs(k1))):
        want = np.array([oracle.solve_node_problem(pt, rhom[i], a[i], Qm[i], rh0[i], b[i], rh1[i], c[i], prefer)
                         for i in range(n)])
        got = ca.node_problem_batch(pt, rhom, a, Qm, rh0, b, rh1, c, prefer=prefer)
        assert np.array_equal(got.view(np.int64), want.view(np.int64)), "pt=%d" % pt
