"""Randomised inputs for the cedr::local solvers, as reference cedr_local.cpp:99-285 generates them
(plus infeasible and degenerate ones), shared by the host and the device parity tests."""
import numpy as np


def bc_qp_cases(n, count, seed):
    """[(w, a, b, xlo, xhi, y)] for 'count' base problems x 5 variants each."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        a = 0.1 + rng.random(n)
        w = 0.1 + rng.random(n)
        xlo = rng.random(n) - 0.5
        xhi = xlo + rng.random(n)
        al, au = float(np.dot(a, xlo)), float(np.dot(a, xhi))
        t = rng.random()
        b_in = t * al + (1 - t) * au
        tt = rng.random(n)
        y_in = tt * xlo + (1 - tt) * xhi
        y_out = y_in.copy()
        y_out[0::2] = (xlo - rng.random(n))[0::2]
        if n > 1:
            y_out[1::2] = (xhi + rng.random(n))[1::2]
        out.append((w, a, b_in, xlo, xhi, y_in))                      # feasible, y inside
        out.append((w, a, float(np.dot(a, y_in)), xlo, xhi, y_in))    # x = y solves it
        out.append((w, a, b_in, xlo, xhi, y_out))                     # y outside the box
        out.append((w, a, al - 0.1, xlo, xhi, y_in))                  # infeasible below
        out.append((w, a, au, xlo, xhi, y_in))                        # the upper corner
    return out


def nonneg_cases(n, count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        b = rng.random() if i % 7 else -0.1
        w = 0.1 + rng.random(n)
        a = 0.1 + rng.random(n)
        y = rng.random(n) * np.where(rng.random(n) > 0.8, -1, 1)
        out.append((w, a, b, y))
    return out
