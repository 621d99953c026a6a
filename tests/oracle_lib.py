"""ctypes binding of oracle/libcedr_oracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module. The product package compose_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_lp = C.POINTER(C.c_long)


def _dp(a):
    return a.ctypes.data_as(c_dp) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(c_ip) if a is not None else None


def _lp(a):
    return a.ctypes.data_as(c_lp) if a is not None else None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ORACLE_DIR, "libcedr_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.or_last_error.restype = C.c_char_p
    L.or_build_flags.restype = C.c_char_p
    for name in ("or_tree_make_1d", "or_tree_from_arrays", "or_analyze", "or_qlt_create",
                 "or_caas_create", "or_qlt_nodesets"):
        getattr(L, name).restype = C.c_void_p
    L.or_tree_make_1d.argtypes = [C.c_int] * 4
    L.or_tree_from_arrays.argtypes = [C.c_int, C.c_int, c_ip, c_ip, c_ip, c_lp, c_ip]
    L.or_tree_nnodes.argtypes = [C.c_void_p]
    L.or_tree_export.argtypes = [C.c_void_p, c_ip, c_ip, c_ip, c_ip, c_lp]
    L.or_tree_free.argtypes = [C.c_void_p]
    L.or_analyze.argtypes = [C.c_void_p, C.c_int, C.c_int]
    for name in ("or_ns_free", "or_ns_nslots", "or_ns_nnode", "or_ns_nlevels", "or_ns_check_comm"):
        getattr(L, name).argtypes = [C.c_void_p]
    for name in ("or_ns_level_size", "or_ns_level_nme", "or_ns_level_nkids", "or_ns_count_leaves"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_int]
    L.or_ns_nodes.argtypes = [C.c_void_p, c_ip]
    L.or_ns_level_nodes.argtypes = [C.c_void_p, C.c_int, c_ip]
    L.or_ns_level_msgs.argtypes = [C.c_void_p, C.c_int, C.c_int, c_ip]
    L.or_tree_unittest.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.or_solve_1eq_bc_qp_2d.argtypes = [c_dp, c_dp, C.c_double, c_dp, c_dp, c_dp, c_dp, C.c_int, C.c_int]
    L.or_solve_1eq_bc_qp.argtypes = [C.c_int, c_dp, c_dp, C.c_double, c_dp, c_dp, c_dp, c_dp, C.c_int]
    L.or_caas_local.argtypes = [C.c_int, c_dp, C.c_double, c_dp, c_dp, c_dp, c_dp, C.c_int]
    L.or_solve_1eq_nonneg.argtypes = [C.c_int, c_dp, C.c_double, c_dp, c_dp, c_dp, C.c_int]
    L.or_solve_node_problem.argtypes = [C.c_int, C.c_double, c_dp, C.c_double, C.c_double, c_dp,
                                        C.c_double, c_dp, C.c_int, c_dp]
    L.or_qlt_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.or_qlt_free.argtypes = [C.c_void_p]
    L.or_qlt_declare.argtypes = [C.c_void_p, c_ip, C.c_int]
    L.or_qlt_set_threads.argtypes = [C.c_void_p, C.c_int]
    L.or_qlt_nlclcells.argtypes = [C.c_void_p, C.c_int]
    L.or_qlt_nslots.argtypes = [C.c_void_p, C.c_int]
    L.or_qlt_l2rndps.argtypes = [C.c_void_p]
    L.or_qlt_r2lndps.argtypes = [C.c_void_p]
    L.or_qlt_nodesets.argtypes = [C.c_void_p, C.c_int]
    L.or_qlt_owned_glblcells.argtypes = [C.c_void_p, C.c_int, c_lp]
    L.or_qlt_meta.argtypes = [C.c_void_p] + [c_ip] * 8
    L.or_qlt_set.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp]
    L.or_qlt_run.argtypes = [C.c_void_p]
    L.or_qlt_get.argtypes = [C.c_void_p, C.c_int, c_dp]
    L.or_qlt_time_run.argtypes = [C.c_void_p, C.c_int]
    L.or_qlt_time_run.restype = C.c_double
    L.or_caas_create.argtypes = [c_ip, C.c_int, c_ip, C.c_int, C.c_int]
    L.or_caas_free.argtypes = [C.c_void_p]
    L.or_caas_set_threads.argtypes = [C.c_void_p, C.c_int]
    L.or_caas_set.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp]
    L.or_caas_set_Qm_only.argtypes = [C.c_void_p, C.c_int, c_dp]
    L.or_caas_run.argtypes = [C.c_void_p]
    L.or_caas_get.argtypes = [C.c_void_p, C.c_int, c_dp]
    L.or_caas_recv.argtypes = [C.c_void_p, c_dp]
    L.or_caas_time_run.argtypes = [C.c_void_p, c_dp, C.c_int]
    L.or_caas_time_run.restype = C.c_double
    L.or_bfb_allreduce.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_dp, c_dp]
    _LIB = L
    return L


def _check(rc, what):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise RuntimeError("%s: %s" % (what, lib().or_last_error().decode()))
    return rc


class Tree:
    """Caller tree (cedr_tree_caller.hpp:12-24) held by the oracle."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("tree: " + lib().or_last_error().decode())
        self.h = handle

    @staticmethod
    def make_1d(ncells, nranks=1, pseudorandom=False, imbalanced=False):
        return Tree(lib().or_tree_make_1d(ncells, nranks, int(pseudorandom), int(imbalanced)))

    @staticmethod
    def from_arrays(root, kid0, kid1, rank, cellidx, level=None):
        kid0 = np.ascontiguousarray(kid0, np.int32)
        kid1 = np.ascontiguousarray(kid1, np.int32)
        rank = np.ascontiguousarray(rank, np.int32)
        cellidx = np.ascontiguousarray(cellidx, np.int64)
        lv = None if level is None else np.ascontiguousarray(level, np.int32)
        return Tree(lib().or_tree_from_arrays(len(kid0), root, _ip(kid0), _ip(kid1), _ip(rank),
                                              _lp(cellidx), _ip(lv)))

    def export(self):
        n = lib().or_tree_nnodes(self.h)
        root = C.c_int()
        kid0 = np.zeros(n, np.int32); kid1 = np.zeros(n, np.int32)
        rank = np.zeros(n, np.int32); cellidx = np.zeros(n, np.int64)
        lib().or_tree_export(self.h, C.byref(root), _ip(kid0), _ip(kid1), _ip(rank), _lp(cellidx))
        return dict(root=root.value, kid0=kid0, kid1=kid1, rank=rank, cellidx=cellidx)

    def unittest(self, ncells, nranks, ntrial=11):
        return _check(lib().or_tree_unittest(self.h, ncells, nranks, ntrial), "tree unittest")

    def __del__(self):
        try:
            lib().or_tree_free(self.h)
        except Exception:
            pass


def dump_nodesets(ns_handle):
    """NodeSets -> python dict of integer arrays (cedr_tree.hpp:16-97)."""
    L = lib()
    nn = L.or_ns_nnode(ns_handle)
    nodes = np.zeros((nn, 7), np.int32)
    L.or_ns_nodes(ns_handle, _ip(nodes))
    levels = []
    for il in range(L.or_ns_nlevels(ns_handle)):
        nl = L.or_ns_level_size(ns_handle, il)
        ln = np.zeros(nl, np.int32)
        L.or_ns_level_nodes(ns_handle, il, _ip(ln))
        me = np.zeros((L.or_ns_level_nme(ns_handle, il), 3), np.int32)
        kids = np.zeros((L.or_ns_level_nkids(ns_handle, il), 3), np.int32)
        if len(me):
            L.or_ns_level_msgs(ns_handle, il, 0, _ip(me))
        if len(kids):
            L.or_ns_level_msgs(ns_handle, il, 1, _ip(kids))
        levels.append(dict(nodes=ln, me=me, kids=kids))
    return dict(nslots=L.or_ns_nslots(ns_handle), nodes=nodes, levels=levels)


def analyze(tree, ncells, rank):
    h = lib().or_analyze(tree.h, ncells, rank)
    if not h:
        raise RuntimeError("analyze: " + lib().or_last_error().decode())
    try:
        return dump_nodesets(h)
    finally:
        lib().or_ns_free(h)


class OracleQLT:
    def __init__(self, tree, ncells, nranks=1, prefer_mass_con=False):
        self.L = lib()
        self.h = self.L.or_qlt_create(tree.h, ncells, nranks, int(prefer_mass_con))
        if not self.h:
            raise RuntimeError("qlt: " + self.L.or_last_error().decode())
        self.nranks = nranks
        self.nt = 0

    def declare(self, types):
        t = np.ascontiguousarray(types, np.int32)
        _check(self.L.or_qlt_declare(self.h, _ip(t), len(t)), "declare")
        self.nt = len(t)

    def set_threads(self, n):
        self.L.or_qlt_set_threads(self.h, n)

    def nlclcells(self, rank=0):
        return self.L.or_qlt_nlclcells(self.h, rank)

    def nslots(self, rank=0):
        return self.L.or_qlt_nslots(self.h, rank)

    def ndps(self):
        return self.L.or_qlt_l2rndps(self.h), self.L.or_qlt_r2lndps(self.h)

    def nodesets(self, rank=0):
        return dump_nodesets(self.L.or_qlt_nodesets(self.h, rank))

    def owned_glblcells(self, rank=0):
        g = np.zeros(self.nlclcells(rank), np.int64)
        self.L.or_qlt_owned_glblcells(self.h, rank, _lp(g))
        return g

    def meta(self):
        a = [np.zeros(self.nt, np.int32) for _ in range(5)] + [np.zeros(7, np.int32) for _ in range(3)]
        self.L.or_qlt_meta(self.h, *[_ip(x) for x in a])
        keys = ["trcr2prob", "bidx2trcr", "trcr2bidx", "trcr2bl2r", "trcr2br2l",
                "prob2trcrptr", "prob2bl2r", "prob2br2l"]
        return dict(zip(keys, a))

    def set(self, rank, rhom, Qm, Qm_min, Qm_max, Qm_prev):
        arrs = [None if a is None else np.ascontiguousarray(a, np.float64)
                for a in (rhom, Qm, Qm_min, Qm_max, Qm_prev)]
        _check(self.L.or_qlt_set(self.h, rank, *[_dp(a) for a in arrs]), "qlt set")

    def run(self):
        _check(self.L.or_qlt_run(self.h), "qlt run")

    def get(self, rank=0):
        out = np.zeros((self.nt, self.nlclcells(rank)))
        self.L.or_qlt_get(self.h, rank, _dp(out))
        return out

    def time_run(self, nrep):
        return self.L.or_qlt_time_run(self.h, nrep)

    def __del__(self):
        try:
            self.L.or_qlt_free(self.h)
        except Exception:
            pass


class OracleCAAS:
    def __init__(self, nlcl, types, accum=0):
        self.L = lib()
        nl = np.ascontiguousarray(np.atleast_1d(nlcl), np.int32)
        t = np.ascontiguousarray(types, np.int32)
        self.h = self.L.or_caas_create(_ip(nl), len(nl), _ip(t), len(t), accum)
        if not self.h:
            raise RuntimeError("caas: " + self.L.or_last_error().decode())
        self.nl = nl
        self.nt = len(t)

    def set_threads(self, n):
        self.L.or_caas_set_threads(self.h, n)

    def set(self, rank, rhom, Qm, Qm_min, Qm_max, Qm_prev):
        arrs = [None if a is None else np.ascontiguousarray(a, np.float64)
                for a in (rhom, Qm, Qm_min, Qm_max, Qm_prev)]
        self.L.or_caas_set(self.h, rank, *[_dp(a) for a in arrs])

    def run(self):
        _check(self.L.or_caas_run(self.h), "caas run")

    def get(self, rank=0):
        out = np.zeros((self.nt, int(self.nl[rank])))
        self.L.or_caas_get(self.h, rank, _dp(out))
        return out

    def recv(self):
        out = np.zeros(4 * self.nt)
        self.L.or_caas_recv(self.h, _dp(out))
        return out

    def time_run(self, Qm, nrep):
        Qm = np.ascontiguousarray(Qm, np.float64)
        return self.L.or_caas_time_run(self.h, _dp(Qm), nrep)

    def __del__(self):
        try:
            self.L.or_caas_free(self.h)
        except Exception:
            pass


def bfb_allreduce(tree, ncells, nranks, nfield, transpose, send_per_rank):
    send = np.ascontiguousarray(np.concatenate([np.ravel(s) for s in send_per_rank]), np.float64)
    recv = np.zeros((nranks, nfield))
    _check(lib().or_bfb_allreduce(tree.h, ncells, nranks, nfield, int(transpose), _dp(send), _dp(recv)),
           "bfb")
    return recv


def solve_2d(w, a, b, xlo, xhi, y, clip=True, early=True):
    arrs = [np.ascontiguousarray(v, np.float64) for v in (w, a, xlo, xhi, y)]
    x = np.zeros(2)
    info = lib().or_solve_1eq_bc_qp_2d(_dp(arrs[0]), _dp(arrs[1]), b, _dp(arrs[2]), _dp(arrs[3]),
                                       _dp(arrs[4]), _dp(x), int(clip), int(early))
    return info, x


def solve_nd(w, a, b, xlo, xhi, y, max_its=100):
    arrs = [np.ascontiguousarray(v, np.float64) for v in (w, a, xlo, xhi, y)]
    n = len(arrs[0])
    x = np.zeros(n)
    info = lib().or_solve_1eq_bc_qp(n, _dp(arrs[0]), _dp(arrs[1]), b, _dp(arrs[2]), _dp(arrs[3]),
                                    _dp(arrs[4]), _dp(x), max_its)
    return info, x


def caas_local(a, b, xlo, xhi, y, clip=True):
    arrs = [np.ascontiguousarray(v, np.float64) for v in (a, xlo, xhi, y)]
    n = len(arrs[0])
    x = np.zeros(n)
    lib().or_caas_local(n, _dp(arrs[0]), b, _dp(arrs[1]), _dp(arrs[2]), _dp(arrs[3]), _dp(x), int(clip))
    return x


def solve_nonneg(a, b, y, w, method, x0=None):
    arrs = [np.ascontiguousarray(v, np.float64) for v in (a, y, w)]
    n = len(arrs[0])
    x = np.zeros(n) if x0 is None else np.array(x0, np.float64)
    info = lib().or_solve_1eq_nonneg(n, _dp(arrs[0]), b, _dp(arrs[1]), _dp(x), _dp(arrs[2]), method)
    return info, x


def solve_node_problem(pt, rhom, pd, Qm, rhom0, k0d, rhom1, k1d, prefer=False):
    pd = np.ascontiguousarray(pd, np.float64)
    k0d = np.ascontiguousarray(k0d, np.float64)
    k1d = np.ascontiguousarray(k1d, np.float64)
    out = np.zeros(2)
    lib().or_solve_node_problem(pt, rhom, _dp(pd), Qm, rhom0, _dp(k0d), rhom1, _dp(k1d), int(prefer),
                                _dp(out))
    return out
