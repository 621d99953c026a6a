"""One-off soak: many seeds of the 36-tracer harness, GPU vs oracle bit for bit."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import compose_amd as ca, oracle_lib as o, randomized as R, common
nseed = int(sys.argv[1]) if len(sys.argv) > 1 else 100
trs = R.tracers_vector(); types = [t.problem_type for t in trs]
bad = 0; n = 0
for nc, imb in ((1000, False), (4099, True), (333, False)):
    for prefer in (False, True):
        q = common.make_hip_qlt(ca, ca.Tree.make_1d(nc, 1, False, imb), nc, types, prefer)
        for seed in range(nseed):
            v = R.make_values(trs, nc, 10000 + seed)
            want = common.run_oracle_qlt(o, (False, imb), nc, 1, types, v, prefer)
            got = common.run_hip_qlt(ca, q, v)
            n += 1
            if not np.array_equal(got, want):
                bad += 1
                print("MISMATCH nc %d imb %s prefer %s seed %d: %d values" % (nc, imb, prefer, seed, (got != want).sum()), flush=True)
    print("nc", nc, "done", flush=True)
print("%d runs, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
