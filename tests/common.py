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


def default_knobs():
    """True unless the environment switches some code path of the library (the env matrix of tools/env_matrix.sh runs the
    suites under such settings): then results must still be right, but which path produced them -- fused or not, how a plan
    is shaped -- is not asserted."""
    import os
    return not any(k.startswith("CEDR_HIP_") and k not in ("CEDR_HIP_PLACE_TRIES", "CEDR_HIP_LIB") for k in os.environ)


def caas_bfb_expected(oracle, v, types, ngroups, accum, imbalanced=False):
    """CAAS paired with the BFB tree reducer over groups of `accum` cells, restated from the reference with the
    ORACLE's tree sum in the middle: clip and per-group partial sums 0 + c0 + c1 + ... (cedr_caas.cpp:140-168),
    oracle.bfb_allreduce over the transposed (field, group) layout (cedr_bfb_tree_allreduce.cpp:79-159),
    adjust_Qm (cedr_caas.cpp:217-252). numpy evaluates every step in IEEE double without fusing, so the HIP
    result must equal this bit for bit."""
    nt, n = v.Qm.shape
    assert n == ngroups * accum
    clip = np.where(v.Qm_max < np.where(v.Qm_min > v.Qm, v.Qm_min, v.Qm), v.Qm_max,
                    np.where(v.Qm_min > v.Qm, v.Qm_min, v.Qm))
    cons = np.array([(t & 1) != 0 for t in types])
    term = np.where(cons[:, None], v.Qm_prev, v.Qm)
    send = np.zeros((4 * nt, ngroups))
    for q, arr in enumerate((clip, term, v.Qm_min, v.Qm_max)):
        a3 = arr.reshape(nt, ngroups, accum)
        acc = np.zeros((nt, ngroups))
        for ai in range(accum):
            acc = acc + a3[:, :, ai]
        send[q * nt:(q + 1) * nt] = acc
    recv = oracle.bfb_allreduce(oracle.Tree.make_1d(ngroups, 1, False, imbalanced), ngroups, 1, 4 * nt, True,
                                [send])[0]
    out = clip.copy()
    for k in range(nt):
        m = recv[nt + k] - recv[k]
        if m < 0:
            fac = recv[k] - recv[2 * nt + k]
            if fac > 0:
                fac = m / fac
                x = out[k] + fac * (out[k] - v.Qm_min[k])
                out[k] = np.where(v.Qm_min[k] > x, v.Qm_min[k], x)
        elif m > 0:
            fac = recv[3 * nt + k] - recv[k]
            if fac > 0:
                fac = m / fac
                x = out[k] + fac * (v.Qm_max[k] - out[k])
                out[k] = np.where(v.Qm_max[k] < x, v.Qm_max[k], x)
    return out


def run_hip_caas_bfb(ca, ngroups, accum, types, v, typed, imbalanced=False, qm_source=False, nrepeat=1):
    """CAAS + BfbTreeAllReducer on one rank. typed: the reducer object itself is CAAS's UserAllReducer
    (caas::BfbUserAllReducer: run() may fuse); else a callback that calls allreduce (never fused).
    Returns (masses, fused?)."""
    p = ca.Parallel.self_()
    ncells = ngroups * accum
    nt = len(types)
    bfb = ca.BfbTreeAllReducer(p, ca.Tree.make_1d(ngroups, 1, False, imbalanced), ngroups, 4 * nt)
    red = bfb if typed else (lambda s, r, nl, nf: bfb.allreduce(s, r, True))
    c = ca.CAAS(p, ncells, red, accum)
    for t in types:
        c.declare_tracer(t, 0)
    c.end_tracer_declarations()
    c.finish_setup()
    d = [ca.DeviceArray.from_numpy(a) for a in (v.rhom, v.Qm, v.Qm_min, v.Qm_max, v.Qm_prev)]
    c.set_rhom(d[0])
    for _ in range(nrepeat):
        c.set_Qm(d[1], d[2], d[3], d[4])
        if qm_source:
            c.set_Qm_source(d[1])
        c.run()
    out_d = ca.DeviceArray(v.Qm.size)
    c.get_Qm(out_d)
    ca.synchronize()
    return out_d.numpy(v.Qm.shape), c.fused_with_bfb()
