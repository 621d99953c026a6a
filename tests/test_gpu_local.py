"""cedr::local on the DEVICE (one problem per GPU thread, cedr_hip_local_batch) against the
oracle, bit for bit: the functions a caller may use inside its own kernels
(reference cedr/cedr_local.hpp:24-60, KOKKOS_INLINE_FUNCTION there)."""
import numpy as np
import pytest

import local_cases as LC

pytestmark = pytest.mark.gpu


def same(x, y):
    return np.array_equal(np.asarray(x).view(np.int64), np.asarray(y).view(np.int64))


def stack(cases):
    w, a, b, xlo, xhi, y = (np.array([c[i] for c in cases]) for i in range(6))
    return w, a, b, xlo, xhi, y


@pytest.mark.parametrize("n", [2, 3, 8, 16])
def test_device_bc_qp_and_caas(gpu, oracle, n):
    ca = gpu
    cases = LC.bc_qp_cases(n, 60, seed=300 + n)
    w, a, b, xlo, xhi, y = stack(cases)
    info, x = ca.local_batch(ca.LOCAL_BC_QP, w, a, b, xlo, xhi, y)
    for i, c in enumerate(cases):
        winfo, wx = oracle.solve_nd(*c)
        assert info[i] == winfo and same(x[i], wx)
    for clip in (True, False):
        _, x = ca.local_batch(ca.LOCAL_CAAS, w, a, b, xlo, xhi, y, clip=clip)
        for i, c in enumerate(cases):
            assert same(x[i], oracle.caas_local(c[1], c[2], c[3], c[4], c[5], clip))


@pytest.mark.parametrize("clip,early", [(True, True), (False, False)])
def test_device_bc_qp_2d(gpu, oracle, clip, early):
    ca = gpu
    cases = LC.bc_qp_cases(2, 400, seed=9)
    w, a, b, xlo, xhi, y = stack(cases)
    info, x = ca.local_batch(ca.LOCAL_BC_QP_2D, w, a, b, xlo, xhi, y, clip=clip, early_exit_on_tol=early)
    for i, c in enumerate(cases):
        winfo, wx = oracle.solve_2d(*c, clip, early)
        assert info[i] == winfo and same(x[i], wx)


@pytest.mark.parametrize("n", [2, 5])
def test_device_nonneg(gpu, oracle, n):
    ca = gpu
    cases = LC.nonneg_cases(n, 300, seed=400 + n)
    w = np.array([c[0] for c in cases]); a = np.array([c[1] for c in cases])
    b = np.array([c[2] for c in cases]); y = np.array([c[3] for c in cases])
    z = np.zeros_like(a)
    for kind, method in ((ca.LOCAL_NONNEG_LS, 0), (ca.LOCAL_NONNEG_CAAS, 1)):
        info, x = ca.local_batch(kind, w, a, b, z, z, y)
        for i, c in enumerate(cases):
            winfo, wx = oracle.solve_nonneg(c[1], c[2], c[3], c[0], method)
            assert info[i] == winfo and same(x[i], wx)
