"""1-D periodic semi-Lagrangian transport demo of the reference
(cedr_test_1d_transport.cpp): cubic interpolation to departure points, then a CDR
per step. Restated with numpy as TEST INFRASTRUCTURE: it drives either the oracle
or the HIP library through one callback, so both see identical inputs step by step."""
import numpy as np


def cubic_interp_periodic(x, y, xi):
    """cedr_test_1d_transport.cpp:40-96. Returns (yi, dod[n][4])."""
    with np.errstate(invalid="ignore", divide="ignore"):  # both arms of np.where are evaluated
        return _cubic_interp_periodic(x, y, xi)


def _cubic_interp_periodic(x, y, xi):
    nx = len(x)
    nc = nx - 1
    w = x[nc] - x[0]
    xp = np.where((xi >= x[0]) & (xi <= x[nc]), xi, xi - w * np.floor((xi - x[0]) / w))
    ip1 = np.searchsorted(x, xp, side="right")
    ip1 = np.where(ip1 == 0, 1, np.where(ip1 == nx, nx - 1, ip1))
    i = ip1 - 1
    dod = (i[:, None] - 1 + np.arange(4)[None, :] + nc) % nc
    slope = lambda a, b: (y[b] - y[a]) / (x[b] - x[a])
    smid = slope(i, ip1)
    at_start, at_end = i == 0, i == nc - 1
    im1 = np.where(at_start, nc - 1, i - 1)
    a_s = (x[nc] - x[nc - 1]) / ((x[1] - x[0]) + (x[nc] - x[nc - 1]))
    a_i = (x[i] - x[np.maximum(i - 1, 0)]) / (x[ip1] - x[np.maximum(i - 1, 0)])
    s1 = np.where(at_start, (1 - a_s) * slope(nc - 1, nc) + a_s * smid,
                  (1 - a_i) * slope(im1, np.where(at_start, nc, i)) + a_i * smid)
    ip2 = np.minimum(i + 2, nc)
    a_e = (x[ip1] - x[i]) / ((x[ip1] - x[i]) + (x[1] - x[0]))
    a_j = (x[ip1] - x[i]) / (x[ip2] - x[i])
    s2 = np.where(at_end, (1 - a_e) * smid + a_e * slope(0, 1),
                  (1 - a_j) * smid + a_j * slope(ip1, np.where(at_end, ip1, ip2)))
    dx = x[ip1] - x[i]
    dx2, dx3 = dx * dx, dx * dx * dx
    den = -dx3
    c2, c3 = s1, y[i]
    b1 = y[ip1] - dx * c2 - c3
    b2 = s2 - c2
    c0 = (2.0 * b1 - dx * b2) / den
    c1 = (-3.0 * dx * b1 + dx2 * b2) / den
    xij = xp - x[i]
    return (((c0 * xij + c1) * xij) + c2) * xij + c3, dod


def initial_condition(xcp):
    """cedr_test_1d_transport.cpp:125-133, 294-298: sine outside [0.4, 0.9], box inside."""
    sin = 0.1 + 0.8 * 0.5 * (1 + np.sin(6 * np.pi * xcp))
    rect = np.where((xcp > 0.66) | (xcp < 0.33), 0.0, 1.0)
    return np.where((xcp < 0.4) | (xcp > 0.9), sin, rect)


class Problem1D:
    def __init__(self, ncells):
        self.n = ncells
        self.xb = np.arange(ncells + 1) / ncells
        self.xb[-1] = 1.0
        xc = 0.5 * (self.xb[:-1] + self.xb[1:])
        self.xcp = np.concatenate([xc, [1 + xc[0]]])
        self.area = self.xb[1:] - self.xb[:-1]

    def cycle(self, nsteps, y0, cdr_step):
        """cdr_step(Qm, Qm_min, Qm_max, Qm_prev) -> Qm_new, masses per cell (:239-262, :171-191)."""
        n = self.n
        xcpi = self.xcp - 1.0 / nsteps
        yp = y0.copy()
        for _ in range(nsteps):
            y, dod = cubic_interp_periodic(self.xcp, yp, xcpi)
            d = dod[:n]
            lo, hi = yp[d].min(1), yp[d].max(1)
            Q = cdr_step(y[:n] * self.area, lo * self.area, hi * self.area, yp[:n] * self.area)
            y = np.concatenate([Q / self.area, [0.0]])
            y[n] = y[0]
            yp = y
        return yp
