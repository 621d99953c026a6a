"""Randomised soak of the multi-rank HOST logic on the CPU (no GPU): random caller trees, rank counts,
ownerships and pinned nodes through the numpy plan executor (tests/plan_exec.py) in all three
exchange forms, against the single-rank oracle. Usage: python tools/soak_level_mode.py [nseeds] [first]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import compose_amd as ca  # noqa: E402
import oracle_lib as oracle  # noqa: E402
import random_trees as RT  # noqa: E402
import randomized as R  # noqa: E402
import test_cpu_plan as TP  # noqa: E402


def main():
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    os.environ["CEDR_HIP_TILE_LEAVES"] = "8"
    trs = R.tracers_vector()[:12]
    types = [t.problem_type for t in trs]
    nbad = 0
    for seed in range(first, first + nseeds):
        rng = np.random.default_rng(seed)
        kind = ["merge", "skew", "chain"][int(rng.integers(0, 3))]
        nleaf = int(rng.integers(2, 80 if kind == "chain" else 600))
        P = int(rng.integers(2, min(9, nleaf + 1)))
        how = ["blocks", "scattered"][int(rng.integers(0, 2))]
        pin = [0.0, 0.0, 0.2, 0.6][int(rng.integers(0, 4))]
        os.environ["CEDR_HIP_MICRO"] = str(int(rng.integers(0, 3)))
        root, kid0, kid1, rank, cell = RT.random_tree(rng, nleaf, kind)
        rank = RT.assign_ranks(rng, nleaf, rank, P, how, pin)
        v = R.make_values(trs, nleaf, seed=seed)
        want = RT.oracle_single_rank(oracle, root, kid0, kid1, cell, nleaf, types, v)
        tree = ca.Tree.from_arrays(root, kid0, kid1, rank, cell)
        for mode, partial in ((0 if pin > 0 else 1, False), (2, False), (2, True)):
            os.environ["CEDR_HIP_EXCHANGE"] = str(mode)
            if partial:
                del os.environ["CEDR_HIP_EXCHANGE"]
            try:
                got = TP.run_tree_plan(ca, nleaf, P, tree, types, v, mode, partial)
                ok = np.array_equal(got, want)
                why = "" if ok else "results differ"
            except ca.CedrError as e:  # orders the planner rejects by design are reported, not counted
                ok, why = True, "rejected: " + str(e).splitlines()[-1][:90]
            except AssertionError as e:
                ok, why = False, "assert: " + str(e)[:200]
            if why:
                print("seed %d %s nleaf %d P %d %s pin %.1f mode %d partial %d: %s" %
                      (seed, kind, nleaf, P, how, pin, mode, partial, why), flush=True)
            nbad += not ok
    print("SOAK %s: %d seeds, %d failures" % ("FAIL" if nbad else "OK", nseeds, nbad))
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main())
