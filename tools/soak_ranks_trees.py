"""One-off soak on the GPU: QLT on P thread-ranks over random caller trees, ownerships, pinned nodes and
exchange forms (all-gather / level mode / partial trees) vs the single-rank oracle, bit for bit.
Usage: python tools/soak_ranks_trees.py [ncases] [seconds]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401  (one librccl in the process, see tests/conftest.py)
import compose_amd as ca, oracle_lib as o, randomized as R, random_trees as RT, threaded_ranks as T
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 60
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 1e9
rng = np.random.default_rng(11)
trs = R.tracers_vector(); types = [t.problem_type for t in trs]
bad = rejected = done = 0
t0 = time.time()
for case in range(ncase):
    if time.time() - t0 > budget:
        break
    kind = ["merge", "skew", "chain"][int(rng.integers(0, 3))]
    nleaf = int(rng.integers(2, 300 if kind == "chain" else 6000))
    P = int(rng.integers(2, min(9, nleaf + 1)))
    how = ["blocks", "scattered"][int(rng.integers(0, 2))]
    pin = [0.0, 0.0, 0.2, 0.6][int(rng.integers(0, 4))]
    prefer = bool(rng.integers(0, 2))
    root, kid0, kid1, rank, cell = RT.random_tree(rng, nleaf, kind)
    rank = RT.assign_ranks(rng, nleaf, rank, P, how, pin)
    v = R.make_values(trs, nleaf, int(rng.integers(0, 1 << 30)))
    want = RT.oracle_single_rank(o, root, kid0, kid1, cell, nleaf, types, v, prefer)
    for mode, partial in ((0 if pin > 0 else 1, False), (2, False), (2, True)):
        os.environ["CEDR_HIP_EXCHANGE"] = str(mode)
        if partial:
            del os.environ["CEDR_HIP_EXCHANGE"]
        out = np.zeros_like(v.Qm)

        def body(r, comm):
            tree = ca.Tree.from_arrays(root, kid0, kid1, rank, cell)
            if partial:
                tree = tree.prune(r)
            q = ca.QLT(comm.parallel(ca, r), nleaf, tree, prefer)
            for t in types:
                q.declare_tracer(t, 0)
            q.end_tracer_declarations()
            q.finish_setup()
            g = q.get_owned_glblcells()
            d = [ca.DeviceArray.from_numpy(a) for a in (v.rhom[g], v.Qm[:, g], v.Qm_min[:, g], v.Qm_max[:, g], v.Qm_prev[:, g])]
            q.set_rhom(d[0]); q.set_Qm(d[1], d[2], d[3], d[4]); q.run()
            od = ca.DeviceArray(len(types) * len(g)); q.get_Qm(od); ca.synchronize()
            out[:, g] = od.numpy((len(types), len(g)))
        try:
            T.run_ranks(P, body, timeout=120)
        except ca.CedrError as e:
            if "not supported yet" in str(e):
                rejected += 1; continue
            raise
        if not np.array_equal(out, want):
            bad += 1
            print("MISMATCH %s nleaf %d P %d %s pin %.1f prefer %s mode %d partial %d: %d values" %
                  (kind, nleaf, P, how, pin, prefer, mode, partial, (out != want).sum()), flush=True)
    done += 1
    if done % 20 == 0: print("case", done, "%.0f s" % (time.time() - t0), flush=True)
print("%d cases x 3 forms, %d mismatches, %d rejected orderings" % (done, bad, rejected))
sys.exit(1 if bad else 0)
