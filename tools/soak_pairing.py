"""Soak: the production pairing -- CAAS fused with the library's BFB tree reducer (caas::BfbUserAllReducer) -- on random
reducer-tree sizes, cells per leaf (every group size from 1 to 12, and 16, 20, 32: register form, LDS form, loop form, not
fusable), balanced and imbalanced trees, random tracer subsets of the harness list, with and without set_Qm_source: bit for
bit against the restatement with the ORACLE's tree sum in the middle (tests/common.py caas_bfb_expected) and against the
same pairing through a callback (the two separate launches). Usage: soak_pairing.py [ncases]."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, compose_amd as ca, common, oracle_lib as o, randomized as R
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(20261005)
ctrs_all = R.caas_tracers(R.tracers_vector())
bad, nfused, t0 = 0, 0, time.time()
for case in range(ncases):
    accum = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16, 20, 32]))
    ngroups = int(rng.integers(2, 30000 // max(accum // 4, 1)))
    imb = bool(rng.integers(0, 2))
    k = int(rng.integers(1, len(ctrs_all) + 1))
    sel = sorted(rng.choice(len(ctrs_all), size=k, replace=False).tolist())
    trs = [R.Tracer(i, ctrs_all[j].problem_type, ctrs_all[j].perturbation_type) for i, j in enumerate(sel)]
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, ngroups * accum, seed=1000 + case)
    want = common.caas_bfb_expected(o, v, types, ngroups, accum, imb)
    qs = bool(rng.integers(0, 2))
    got, fused = common.run_hip_caas_bfb(ca, ngroups, accum, types, v, typed=True, imbalanced=imb, qm_source=qs,
                                         nrepeat=int(rng.integers(1, 3)))
    cb, f0 = common.run_hip_caas_bfb(ca, ngroups, accum, types, v, typed=False, imbalanced=imb)
    nfused += fused
    if f0 or not np.array_equal(got, want) or not np.array_equal(cb, want):
        bad += 1
        print("MISMATCH case %d: ngroups %d accum %d imbalanced %s nt %d fused %s" % (case, ngroups, accum, imb, k, fused), flush=True)
    if case % 20 == 19:
        print("  %d cases, %d fused, %d mismatches, %.0f s" % (case + 1, nfused, bad, time.time() - t0), flush=True)
print("soak_pairing: %d cases (%d fused), %d mismatches" % (ncases, nfused, bad))
sys.exit(1 if bad else 0)
