"""f2: host setup at scale. tree::analyze (+ the tile plan) of the product against the oracle's analyze
(the reference's algorithm: shared_ptr DFS + stable_sort per level, cedr_tree.cpp:55-213) on the same host,
single rank, and the integers compared. Usage: setup_time.py [ncells ...]   (CPU only)"""
import os, sys, time, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import compose_amd as ca, oracle_lib as o

def t(f):
    s = time.perf_counter(); r = f(); return r, time.perf_counter() - s

for n in [int(x) for x in (sys.argv[1:] or ["1500000", "6000000"])]:
    L, OL = ca.lib(), o.lib()
    tr, t_tree = t(lambda: ca.Tree.make_1d(n))
    h, t_an = t(lambda: L.cedr_tree_analyze(tr.h, n, 0))
    ph, t_plan = t(lambda: L.cedr_plan_create(tr.h, n, 0, 1, 0))
    otr, t_otree = t(lambda: o.Tree.make_1d(n))
    oh, t_oan = t(lambda: OL.or_analyze(otr.h, n, 0))
    # identical integers: slots, node table (rank, id, parent, nkids, kids, offset), level lists
    assert L.cedr_ns_nslots(h) == OL.or_ns_nslots(oh) and L.cedr_ns_nnode(h) == OL.or_ns_nnode(oh)
    nn = L.cedr_ns_nnode(h)
    a = np.zeros((nn, 7), np.int32); b = np.zeros((nn, 7), np.int32)
    L.cedr_ns_nodes(h, a.ctypes.data_as(C.POINTER(C.c_int))); OL.or_ns_nodes(oh, b.ctypes.data_as(C.POINTER(C.c_int)))
    same = bool((a == b).all()) and L.cedr_ns_nlevels(h) == OL.or_ns_nlevels(oh)
    for il in range(L.cedr_ns_nlevels(h)):
        m = L.cedr_ns_level_size(h, il)
        same = same and m == OL.or_ns_level_size(oh, il)
        x = np.zeros(m, np.int32); y = np.zeros(m, np.int32)
        L.cedr_ns_level_nodes(h, il, x.ctypes.data_as(C.POINTER(C.c_int))); OL.or_ns_level_nodes(oh, il, y.ctypes.data_as(C.POINTER(C.c_int)))
        same = same and bool((x == y).all())
    print("%9d cells: product make_tree %.2f s, analyze %.2f s, plan (incl. its own analyze) %.2f s | oracle make_tree %.2f s, "
          "analyze %.2f s | identical integers: %s" % (n, t_tree, t_an, t_plan, t_otree, t_oan, same))
    L.cedr_ns_free(h); L.cedr_plan_free(ph); OL.or_ns_free(oh)
    del tr, otr
