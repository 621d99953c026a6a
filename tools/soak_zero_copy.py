"""One-off soak: zero-copy (bind_arrays) vs packed path over random sizes and trees, bit for bit."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import compose_amd as ca, randomized as R, common
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(2026)
trs = R.tracers_vector(); types = [t.problem_type for t in trs]
bad = 0
for case in range(ncase):
    nc = int(rng.integers(1, 700)) if case % 3 else int(rng.integers(700, 6000))
    imb = bool(rng.integers(0, 2)); prefer = bool(rng.integers(0, 2))
    v = R.make_values(trs, nc, int(rng.integers(0, 1 << 30)))
    q = common.make_hip_qlt(ca, ca.Tree.make_1d(nc, 1, False, imb), nc, types, prefer)
    want = common.run_hip_qlt(ca, q, v)
    d = [ca.DeviceArray.from_numpy(a) for a in (v.rhom, v.Qm, v.Qm_min, v.Qm_max, v.Qm_prev)]
    q.bind_arrays(d[0], d[1], d[2], d[3], d[4], d[1])  # in place
    q.run(); ca.synchronize()
    got = d[1].numpy(v.Qm.shape)
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH nc %d imb %s prefer %s: %d values" % (nc, imb, prefer, (got != want).sum()), flush=True)
    # CAAS
    ct = R.caas_tracers(trs); ctypes_ = [t.problem_type for t in ct]
    vc = R.make_values(ct, nc, case)
    pk = common.run_hip_caas(ca, nc, ctypes_, vc)
    c = ca.CAAS(ca.Parallel.self_(), nc)
    for t in ctypes_: c.declare_tracer(t, 0)
    c.end_tracer_declarations(); c.finish_setup()
    dc = [ca.DeviceArray.from_numpy(a) for a in (vc.Qm, vc.Qm_min, vc.Qm_max, vc.Qm_prev)]
    c.bind_arrays(dc[0], dc[1], dc[2], dc[3]); c.run(); ca.synchronize()
    if not np.array_equal(dc[0].numpy(vc.Qm.shape), pk):
        bad += 1
        print("CAAS MISMATCH nc %d" % nc, flush=True)
    if case % 50 == 49: print("case", case + 1, flush=True)
print("%d cases, %d mismatches" % (ncase, bad))
sys.exit(1 if bad else 0)
