"""local::unittest (reference cedr_local.cpp:99-294) against the oracle: randomized
1-equality bound-constrained QPs checked with the first-order optimality conditions
(:12-97), the 2-D special case against the N-D solver, local caas, nonneg wrappers."""
import numpy as np

EPS = np.finfo(float).eps


def urand(rng, n=None):
    return rng.random(n)


def calc_r_tol(b, a, y):
    return 10 * EPS * max(abs(b), np.abs(a * y).max())


def check_foc(w, a, b, xlo, xhi, y, x):
    """cedr_local.cpp:12-97"""
    ok = not ((x < xlo).any() or (x > xhi).any())
    r = float(np.dot(a, x)) - b
    ok = ok and abs(r) <= calc_r_tol(b, a, y) * (1 + 1e-3) + 4 * EPS * np.abs(a * x).sum()
    pad = 1e5 * EPS * (xhi - xlo)
    inside = (xlo + pad <= x) & (x <= xhi - pad)
    lam = 0.0
    if inside.any():
        lam = np.dot(a[inside], (w * (x - y))[inside]) / np.dot(a[inside], a[inside])
    g = w * (x - y) - a * lam
    gtol = 1e4 * EPS * np.linalg.norm(y)
    ok = ok and np.linalg.norm(g[inside]) <= gtol
    onlo = x == xlo
    onhi = (~onlo) & (x == xhi)
    ok = ok and not (g[onlo] < -gtol).any() and not (g[onhi] > gtol).any()
    return ok


def test_1eq_bc_qp(oracle):  # cedr_local.cpp:99-222
    rng = np.random.default_rng(1)
    nerr = 0
    for n in range(2, 17):
        for _ in range(100 if n == 2 else 10):
            a = 0.1 + urand(rng, n)
            w = 0.1 + urand(rng, n)
            xlo = urand(rng, n) - 0.5
            xhi = xlo + urand(rng, n)
            al, au = np.dot(a, xlo), np.dot(a, xhi)

            def genb_in():
                t = urand(rng)
                return t * al + (1 - t) * au

            def geny(inside):
                if inside:
                    t = urand(rng, n)
                    return t * xlo + (1 - t) * xhi
                y = np.zeros(n)
                if urand(rng) > 0.2:
                    t = urand(rng, n)
                    y[1::2] = (t * xlo + (1 - t) * xhi)[1::2]
                    y[0::4] = (xlo - urand(rng, n))[0::4]
                    y[2::4] = (xhi + urand(rng, n))[2::4]
                else:
                    y[0::2] = (xlo - urand(rng, n))[0::2]
                    y[1::2] = (xhi + urand(rng, n))[1::2]
                return y

            cases = []
            b, y = genb_in(), geny(True)
            cases.append((b, y))
            cases.append((float(np.dot(a, y)), y))
            cases.append((genb_in(), geny(False)))
            for b, y in cases:
                info, x = oracle.solve_nd(w, a, b, xlo, xhi, y)
                assert info in (0, 1)
                if not check_foc(w, a, b, xlo, xhi, y, x):
                    nerr += 1
                if n == 2:
                    info2, x2 = oracle.solve_2d(w, a, b, xlo, xhi, y)
                    assert info2 == 1
                    rd = np.abs(x - x2).sum() / np.abs(x).sum()
                    assert rd <= 1e4 * EPS
                xc = oracle.caas_local(a, b, xlo, xhi, y)
                assert (xc >= xlo).all() and (xc <= xhi).all()
                assert abs(b - np.dot(a, xc)) / np.abs(a * xc).sum() <= 1e3 * EPS
    assert nerr == 0


def test_1eq_nonneg(oracle):  # cedr_local.cpp:224-285
    rng = np.random.default_rng(2)
    for _ in range(200):
        b = 0.5 * 2 * urand(rng)
        w = 0.1 + urand(rng, 2)
        a = 0.1 + urand(rng, 2)
        xhi = b / a
        y = urand(rng, 2) * np.where(urand(rng, 2) > 0.8, -1, 1)
        _, x1_caas = oracle.solve_nonneg(a, b, y, w, 1)
        x_caas = oracle.caas_local(a, b, np.zeros(2), xhi, y)
        _, x1_ls = oracle.solve_nonneg(a, b, y, w, 0)
        _, x_ls = oracle.solve_nd(w, a, b, np.zeros(2), xhi, y)
        rd = lambda p, q: np.abs(p - q).sum() / max(np.abs(p).sum(), 1e-300)
        assert rd(x_caas, x1_caas) <= 10 * EPS
        assert rd(x_ls, x1_ls) <= 50 * EPS
    # infeasible: b < 0 leaves x untouched and returns -1 (cedr_local_inl.hpp:312)
    info, x = oracle.solve_nonneg([1, 1], -1.0, [0.2, 0.3], [1, 1], 0, x0=[7, 8])
    assert info == -1 and list(x) == [7, 8]


def test_infeasible_2d_returns_corner(oracle):
    """check_lu (cedr_local_inl.hpp:23-41): infeasible problems return -1 with x at a corner."""
    info, x = oracle.solve_2d([1, 1], [1, 1], 10.0, [0, 0], [1, 1], [0.5, 0.5])
    assert info == -1 and list(x) == [1, 1]
    info, x = oracle.solve_2d([1, 1], [1, 1], -10.0, [0, 0], [1, 1], [0.5, 0.5])
    assert info == -1 and list(x) == [0, 0]
