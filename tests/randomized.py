"""Python restatement of the reference's randomized property-test harness
(cedr_test_randomized.{hpp,cpp,_inl.hpp}): the 36-tracer table, the input
generators, the six perturbations and check() with the reference's tolerances.

It drives any CDR through a callback `run(vals) -> Qm_out[nt][ncells]` on GLOBAL
arrays indexed by global cell id; rank partitioning is the callback's business.
PRNG: numpy's, seeded by the caller, quantised to multiples of 2^-31 exactly
like the reference's urand() = rand()/(RAND_MAX+1.0) (cedr_test_randomized.cpp:53).
The quantisation matters: the 'nn' tracers' no_change_should_hold check only
passes because sums of 31-bit fractions are exact in binary64.
"""
import numpy as np

CONSERVE, SHAPEPRESERVE, CONSISTENT, NONNEGATIVE = 1, 2, 4, 8
EPS = np.finfo(np.float64).eps

# cedr_test_randomized.cpp:26-33
PTS = [CONSERVE | SHAPEPRESERVE | CONSISTENT, SHAPEPRESERVE, CONSERVE | CONSISTENT,
       CONSISTENT, NONNEGATIVE, NONNEGATIVE | CONSERVE]


def urand(rng, n=None):
    """Uniform in [0,1) on the 2^-31 grid, like rand()/(RAND_MAX+1.0)."""
    if n is None:
        return float(rng.integers(0, 2**31)) / 2.0**31
    return rng.integers(0, 2**31, size=n).astype(np.float64) / 2.0**31


class Tracer:
    def __init__(self, idx, problem_type, perturb):
        self.idx = idx
        self.problem_type = problem_type
        self.perturbation_type = perturb
        sp = bool(problem_type & SHAPEPRESERVE)
        nn = bool(problem_type & NONNEGATIVE)
        self.no_change_should_hold = perturb == 0
        self.safe_should_hold = True
        self.local_should_hold = perturb < 4 and (sp or nn)

    def __repr__(self):
        s = "(ti %d" % self.idx
        for bit, nm in ((CONSERVE, " c"), (SHAPEPRESERVE, " s"), (CONSISTENT, " t"), (NONNEGATIVE, " nn")):
            if self.problem_type & bit:
                s += nm
        return s + " pt %d lsh %d)" % (self.perturbation_type, self.local_should_hold)


def tracers_vector():
    """cedr_test_randomized.cpp:24-51: 6 perturbations x 6 problem types."""
    out = []
    for perturb in range(6):
        for pt in PTS:
            out.append(Tracer(len(out), pt, perturb))
    return out


def caas_tracers(tracers):
    """cedr_caas.cpp:338-351: CAAS keeps shapepreserve && local_should_hold."""
    keep = []
    for t in tracers:
        if not (t.problem_type & SHAPEPRESERVE) or not t.local_should_hold:
            continue
        u = Tracer(len(keep), t.problem_type, t.perturbation_type)
        keep.append(u)
    return keep


class Values:
    def __init__(self, nt, ncells):
        self.ncells = ncells
        self.rhom = np.zeros(ncells)
        self.Qm_min = np.zeros((nt, ncells))
        self.Qm = np.zeros((nt, ncells))
        self.Qm_max = np.zeros((nt, ncells))
        self.Qm_prev = np.zeros((nt, ncells))


def generate_rho(v, rng):  # :55-60
    v.rhom[:] = 0.5 * (1 + urand(rng, v.ncells))


def generate_Q(t, v, rng):  # :62-93
    n = v.ncells
    i = t.idx
    if t.problem_type & NONNEGATIVE:
        if t.no_change_should_hold:
            v.Qm[i] = urand(rng, n)
        else:
            v.Qm[i] = np.where(np.arange(n) % 2 == 0, 0.75, -0.75) + urand(rng, n)
        v.Qm_min[i] = 0
        v.Qm_max[i] = 10
    else:
        q_min = -0.75 + urand(rng, n)
        q_max = q_min + urand(rng, n)
        q = q_min + (q_max - q_min) * urand(rng, n)
        v.Qm_min[i] = q_min * v.rhom
        v.Qm_max[i] = q_max * v.rhom
        v.Qm[i] = np.maximum(v.Qm_min[i], np.minimum(v.Qm_max[i], q * v.rhom))
    v.Qm_prev[i] = v.Qm[i]


def permute_Q(t, v, rng):  # :95-118
    n = v.ncells
    p = np.arange(n)
    for _ in range(n):
        j, k = int(urand(rng) * n), int(urand(rng) * n)
        p[j], p[k] = p[k], p[j]
    v.Qm[t.idx] = v.Qm[t.idx][p]


def add_const_to_Q(t, v, rng, alpha, conserve_mass, safety_problem):  # :120-173
    i = t.idx
    n = v.ncells
    rhom, Qm, Qm_max = v.rhom.sum(), v.Qm[i].sum(), v.Qm_max[i].sum()
    Qm_max_safety = 0.0
    if safety_problem and n:
        Qm_max_safety = np.max(v.Qm_max[i] / v.rhom) * rhom
    if safety_problem:
        dQm = ((Qm_max - Qm) + alpha * (Qm_max_safety - Qm_max)) / n
    else:
        dQm = alpha * (Qm_max - Qm) / n
    v.Qm[i] += dQm
    permute_Q(t, v, rng)
    relax = 0.9
    if conserve_mass:
        dprev = dQm
    elif safety_problem:
        dprev = ((Qm_max - Qm) + relax * alpha * (Qm_max_safety - Qm_max)) / n
    else:
        dprev = relax * alpha * (Qm_max - Qm) / n
    v.Qm_prev[i] += dprev


def perturb_Q(t, v, rng):  # :175-197
    cm = not (t.problem_type & CONSERVE)
    edg = 1 - v.ncells * EPS
    p = t.perturbation_type
    if p == 1:
        permute_Q(t, v, rng)
    elif p == 2:
        add_const_to_Q(t, v, rng, 0.5, cm, False)
    elif p == 3:
        add_const_to_Q(t, v, rng, edg, cm, False)
    elif p == 4:
        add_const_to_Q(t, v, rng, 0.5, cm, True)
    elif p == 5:
        add_const_to_Q(t, v, rng, edg, cm, True)


def make_values(tracers, ncells, seed):
    rng = np.random.default_rng(seed)
    v = Values(len(tracers), ncells)
    generate_rho(v, rng)
    for t in tracers:
        generate_Q(t, v, rng)
        perturb_Q(t, v, rng)
    return v


def check(tracers, v, Qm_out, prefer_mass_con=False, verbose=True):
    """cedr_test_randomized.cpp:293-418. Returns (nerr, messages)."""
    ulp3 = 3 * EPS
    lv_tol = 100 * EPS if prefer_mass_con else 0.0
    safety_tol = 100 * EPS if prefer_mass_con else ulp3
    nerr = 0
    msgs = []
    for t in tracers:
        i = t.idx
        Qm, lo, hi, prev, rhom = Qm_out[i], v.Qm_min[i], v.Qm_max[i], v.Qm_prev[i], v.rhom
        nn = bool(t.problem_type & NONNEGATIVE)
        safe_only = not t.local_should_hold
        lv = (Qm < 0) if nn else ((Qm < lo - lv_tol) | (Qm > hi + lv_tol))
        t_ok = True
        if not safe_only and lv.any():
            k = int(np.argmax(lv))
            msgs.append("check q %r: cell %d Qm %.17g [%.17g, %.17g]" % (t, k, Qm[k], lo[k], hi[k]))
            nerr += int(lv.sum())
            t_ok = False
        if t.no_change_should_hold and (Qm != prev).any():
            k = int(np.argmax(Qm != prev))
            msgs.append("Q should be unchanged but is not %r: %.17g -> %.17g" % (t, prev[k], Qm[k]))
            nerr += int((Qm != prev).sum())
            t_ok = False
        if safe_only:
            q_min = 0.0 if nn else np.min(lo / rhom)
            q_max = np.max(hi / rhom)
            delta = (q_max - q_min) * safety_tol
            bad = (Qm < -ulp3) if nn else ((Qm < q_min * rhom - delta) | (Qm > q_max * rhom + delta))
            if bad.any():
                k = int(np.argmax(bad))
                msgs.append("check q (safety) %r: cell %d %.17g" % (t, k, Qm[k]))
                nerr += int(bad.sum())
                t_ok = False
        desired, actual, den = prev.sum(), Qm.sum(), np.abs(prev).sum()
        rd = abs(actual - desired) / abs(den)
        if rd > 1e2 * EPS:
            msgs.append("FAIL %r mass re %.3e" % (t, rd))
            nerr += 1
    if verbose and msgs:
        print("\n".join(msgs[:20]))
    return nerr, msgs
