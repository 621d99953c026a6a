"""Shared drivers for the parity tests: run the same global problem through the
oracle and through the HIP library (via the C ABI) and hand back global arrays."""
import numpy as np


def run_oracle_qlt(o, tree_args, ncells, nranks, types, v, prefer=False):
    tree = o.Tree.make_1d(ncells, nranks, *tree_args)
    q = o.OracleQLT(tree, ncells, nranks, prefer)
    q.declare(types)
    for r in range(nranks):
        g = q.owned_glblcells(r)
        q.set(r, v.rhom[g], v.Qm[:, g], v.Qm_min[:, g], v.Qm_max[:, g], v.Qm_prev[:, g])
    q.run()
    out = np.zeros_like(v.Qm)
    for r in range(nranks):
        out[:, q.owned_glblcells(r)] = q.get(r)
    return out


def make_hip_qlt(ca, tree, ncells, types, prefer=False, external_memory=False, parallel=None):
    p = parallel or ca.Parallel.self_()
    q = ca.QLT(p, ncells, tree, prefer)
    for t in types:
        q.declare_tracer(t, 0)
    q.end_tracer_declarations()
    if external_memory:
        b1, b2 = q.get_buffers_sizes()
        q.set_buffers(ca.DeviceArray(b1), ca.DeviceArray(b2))
    q.finish_setup()
    return q


def run_hip_qlt(ca, q, v, nrepeat=1):
    """set_rhom, then (set_Qm, run) nrepeat times, then get_Qm -- the call order of
    TestRandomized::run (cedr_test_randomized_inl.hpp:32-58). Single rank."""
    g = q.get_owned_glblcells()
    d = [ca.DeviceArray.from_numpy(a) for a in
         (v.rhom[g], v.Qm[:, g], v.Qm_min[:, g], v.Qm_max[:, g], v.Qm_prev[:, g])]
    q.set_rhom(d[0])
    for _ in range(nrepeat):
        q.set_Qm(d[1], d[2], d[3], d[4])
        q.run()
    out_d = ca.DeviceArray(v.Qm[:, g].size)
    q.get_Qm(out_d)
    ca.synchronize()
    out = np.zeros_like(v.Qm)
    out[:, g] = out_d.numpy(v.Qm[:, g].shape)
    return out


def run_oracle_caas(o, nlcl, types, v, accum=0):
    """nlcl: list of per-rank cell counts; global arrays are split contiguously."""
    c = o.OracleCAAS(nlcl, types, accum)
    start = 0
    for r, n in enumerate(nlcl):
        sl = slice(start, start + n)
        c.set(r, v.rhom[sl], v.Qm[:, sl], v.Qm_min[:, sl], v.Qm_max[:, sl], v.Qm_prev[:, sl])
        start += n
    c.run()
    out = np.zeros_like(v.Qm)
    start = 0
    for r, n in enumerate(nlcl):
        out[:, start:start + n] = c.get(r)
        start += n
    return out, c.recv()


def run_hip_caas(ca, ncells, types, v, external_memory=False, user_reducer=None, accum=1,
                 qm_source=False):
    p = ca.Parallel.self_()
    c = ca.CAAS(p, ncells, user_reducer, accum)
    for t in types:
        c.declare_tracer(t, 0)
    c.end_tracer_declarations()
    if external_memory:
        b1, b2 = c.get_buffers_sizes()
        c.set_buffers(ca.DeviceArray(b1), ca.DeviceArray(b2))
    c.finish_setup()
    d = [ca.DeviceArray.from_numpy(a) for a in (v.rhom, v.Qm, v.Qm_min, v.Qm_max, v.Qm_prev)]
    c.set_rhom(d[0])
    c.set_Qm(d[1], d[2], d[3], d[4])
    if qm_source:
        c.set_Qm_source(d[1])
    c.run()
    out_d = ca.DeviceArray(v.Qm.size)
    c.get_Qm(out_d)
    ca.synchronize()
    return out_d.numpy(v.Qm.shape)


def rel_per_tracer(got, want):
    """max_i |got - want| / max_i |want| for every tracer (row)."""
    return np.abs(got - want).max(axis=1) / np.abs(want).max(axis=1)


def caas_exact_sums(v, tracers):
    """CAAS (reference cedr_caas.cpp:129-253: clip; four sums; redistribute) for the given tracers
    with every sum EXACTLY rounded (math.fsum) -- what a summation of unbounded precision gives.
    Conserving, shape-preserving tracers (the throughput configuration)."""
    import math
    out = np.empty((len(tracers), v.Qm.shape[1]))
    for k, t in enumerate(tracers):
        lo, hi = v.Qm_min[t], v.Qm_max[t]
        clip = np.maximum(lo, np.minimum(hi, v.Qm[t]))
        s_clip, s_prev = math.fsum(clip), math.fsum(v.Qm_prev[t])
        m = s_prev - s_clip
        x = clip.copy()
        if m < 0:
            fac = s_clip - math.fsum(lo)
            if fac > 0:
                x = np.maximum(lo, clip + (m / fac) * (clip - lo))
        elif m > 0:
            fac = math.fsum(hi) - s_clip
            if fac > 0:
                x = np.minimum(hi, clip + (m / fac) * (hi - clip))
        out[k] = x
    return out
