"""The product's public cedr::local solvers (compose_amd/csrc/cedr_local.hpp; reference
cedr/cedr_local.hpp:24-60), evaluated on the HOST through the C ABI, against the oracle's
restatement of cedr_local_inl.hpp: same return value, same bits."""
import numpy as np
import pytest

import local_cases as LC


def same(x, y):
    return np.array_equal(np.asarray(x).view(np.int64), np.asarray(y).view(np.int64))


@pytest.mark.parametrize("n", [2, 3, 5, 16])
def test_solve_1eq_bc_qp(ca, oracle, n):
    for w, a, b, xlo, xhi, y in LC.bc_qp_cases(n, 40, seed=n):
        info, x = ca.local_solve(ca.LOCAL_BC_QP, w, a, b, xlo, xhi, y)
        winfo, wx = oracle.solve_nd(w, a, b, xlo, xhi, y)
        assert info == winfo and same(x, wx)


@pytest.mark.parametrize("clip,early", [(True, True), (False, False), (True, False), (False, True)])
def test_solve_1eq_bc_qp_2d(ca, oracle, clip, early):
    for w, a, b, xlo, xhi, y in LC.bc_qp_cases(2, 150, seed=7):
        info, x = ca.local_solve(ca.LOCAL_BC_QP_2D, w, a, b, xlo, xhi, y, clip=clip, early_exit_on_tol=early)
        winfo, wx = oracle.solve_2d(w, a, b, xlo, xhi, y, clip, early)
        assert info == winfo and same(x, wx)


@pytest.mark.parametrize("n", [1, 2, 7, 16])
def test_caas(ca, oracle, n):
    for w, a, b, xlo, xhi, y in LC.bc_qp_cases(n, 40, seed=100 + n):
        for clip in (True, False):
            _, x = ca.local_solve(ca.LOCAL_CAAS, w, a, b, xlo, xhi, y, clip=clip)
            assert same(x, oracle.caas_local(a, b, xlo, xhi, y, clip))


@pytest.mark.parametrize("n", [2, 4, 16])
def test_solve_1eq_nonneg(ca, oracle, n):
    z = np.zeros(n)
    for w, a, b, y in LC.nonneg_cases(n, 100, seed=200 + n):
        for kind, method in ((ca.LOCAL_NONNEG_LS, 0), (ca.LOCAL_NONNEG_CAAS, 1)):
            info, x = ca.local_solve(kind, w, a, b, z, z, y)
            winfo, wx = oracle.solve_nonneg(a, b, y, w, method)
            assert info == winfo and same(x, wx)


def test_bad_arguments(ca):
    assert ca.local_solve(ca.LOCAL_BC_QP_2D, [1] * 3, [1] * 3, 1.0, [0] * 3, [1] * 3, [0.5] * 3)[0] == -3
    assert ca.local_solve(99, [1, 1], [1, 1], 1.0, [0, 0], [1, 1], [0.5, 0.5])[0] == -3
