"""One-off soak: QLT on P thread-ranks (random P, size, ownership) vs the single-rank oracle, bit for bit."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import compose_amd as ca, oracle_lib as o, randomized as R, common, threaded_ranks as T
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(7)
trs = R.tracers_vector(); types = [t.problem_type for t in trs]
bad = rejected = 0
for case in range(ncase):
    P = int(rng.integers(2, 9))
    nc = int(rng.integers(P, 40)) if case % 4 == 0 else int(rng.integers(P, 3000))
    pr = bool(rng.integers(0, 2)); imb = bool(rng.integers(0, 2)); prefer = bool(rng.integers(0, 2))
    v = R.make_values(trs, nc, int(rng.integers(0, 1 << 30)))
    want = common.run_oracle_qlt(o, (False, imb), nc, 1, types, v, prefer)
    try:
        got = T.run_qlt(ca, P, nc, types, v, prefer, pr, imb)
    except ca.CedrError as e:
        if "not supported yet" in str(e):
            rejected += 1; continue
        raise
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH P %d nc %d pseudorandom %s imbalanced %s prefer %s: %d values" % (P, nc, pr, imb, prefer, (got != want).sum()), flush=True)
    if case % 20 == 19: print("case", case + 1, flush=True)
print("%d cases, %d mismatches, %d rejected orderings" % (ncase, bad, rejected))
sys.exit(1 if bad else 0)
