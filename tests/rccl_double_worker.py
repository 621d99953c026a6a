"""Run by tests/test_gpu_rccl_double.py in its own process with CEDR_HIP_RCCL_LIB = tests/cpp/libfake_rccl.so:
the product's RCCL communicator class (cedr_mpi.cpp RcclComm) with more than one rank -- ranks as threads on
one GPU, the nccl* entry points served by the tests' double. QLT through the all-gather and through the
per-level point-to-point groups, CAAS's all-reduce, the BFB reducer; everything bitwise against the
single-rank oracle. Prints one JSON line with the double's call counts."""
import ctypes
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]

import common  # noqa: E402
import compose_amd as ca  # noqa: E402
import oracle_lib as oracle  # noqa: E402
import randomized as R  # noqa: E402
import threaded_ranks as T  # noqa: E402


def stats(dbl):
    a = (ctypes.c_long * 6)()
    dbl.fake_rccl_stats(a)
    return dict(zip(("allgather", "allreduce", "groups", "sends", "recvs", "errors"), list(a)))


def qlt_case(nranks, ncells, pseudorandom, imbalanced, partial, exchange):
    if exchange:
        os.environ["CEDR_HIP_EXCHANGE"] = str(exchange)
    else:
        os.environ.pop("CEDR_HIP_EXCHANGE", None)
    trs = R.tracers_vector()
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, ncells, seed=31 + ncells + nranks)
    want = common.run_oracle_qlt(oracle, (False, imbalanced), ncells, 1, types, v, False)
    got = T.run_qlt(ca, nranks, ncells, types, v, False, pseudorandom, imbalanced, nrepeat=2, partial=partial,
                    want_level_mode=bool(partial or exchange == 2), comm=T.RcclThreads(ca, nranks, 120))
    assert np.array_equal(got, want), ("QLT", nranks, ncells, partial, exchange)
    assert R.check(trs, v, got, False, verbose=False)[0] == 0


def caas_case(nranks, ncells):
    trs = R.caas_tracers(R.tracers_vector())
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, ncells, seed=77)
    got, nlcl = T.run_caas(ca, nranks, ncells, types, v, comm=T.RcclThreads(ca, nranks, 120))
    want, _ = common.run_oracle_caas(oracle, nlcl, types, v)
    assert common.rel_per_tracer(got, want).max() <= 1e-13


def bfb_case(nranks, ncell, transpose, partial):
    os.environ.pop("CEDR_HIP_EXCHANGE", None)
    nfield = 5
    glob = np.random.default_rng(ncell + nranks).random((ncell, nfield))
    lay = lambda a: np.ascontiguousarray(a.T if transpose else a).ravel()  # noqa: E731
    want = oracle.bfb_allreduce(oracle.Tree.make_1d(ncell, 1, False, False), ncell, 1, nfield, transpose,
                                [lay(glob)])[0]
    got = [None] * nranks

    def body(rank, comm):
        par = comm.parallel(ca, rank)
        tree = ca.Tree.make_1d(ncell, nranks, False, False)
        if partial:
            tree = tree.prune(rank)
        ns = tree.analyze(ncell, rank)
        mine = np.array([ns["nodes"][i][1] for i in ns["levels"][0]["nodes"]], dtype=np.int64)
        b = ca.BfbTreeAllReducer(par, tree, ncell, nfield)
        sd = ca.DeviceArray.from_numpy(lay(glob[mine]))
        rd = ca.DeviceArray(nfield)
        for _ in range(2):
            b.allreduce(sd, rd, transpose)
        ca.synchronize()
        got[rank] = rd.numpy()
    T.run_ranks(nranks, body, comm=T.RcclThreads(ca, nranks, 120))
    for r in range(nranks):
        assert np.array_equal(got[r], want), ("BFB", nranks, ncell, r)


def main():
    path = os.environ.get("CEDR_HIP_RCCL_LIB", "")
    assert path.endswith("libfake_rccl.so") and os.path.exists(path), "CEDR_HIP_RCCL_LIB must name the double"
    dbl = ctypes.CDLL(path)  # the same handle the product binds (same path)
    s0 = stats(dbl)
    # the all-gather of the boundary roots + replicated apex
    for nranks, ncells, pr, imb in [(2, 24, False, False), (3, 111, False, True), (4, 1000, False, False),
                                    (8, 5000, False, False), (8, 8, False, False), (8, 1000, True, False)]:
        qlt_case(nranks, ncells, pr, imb, False, 1 if pr else 0)
    s1 = stats(dbl)
    assert s1["allgather"] > s0["allgather"] and s1["groups"] == s0["groups"], (s0, s1)
    # partial caller trees and forced per-level exchange: groups of sends / receives
    for nranks, ncells, pr, imb, partial in [(2, 24, False, False, True), (3, 111, False, True, True),
                                             (4, 1000, False, False, False), (8, 1000, True, False, False),
                                             (8, 5000, False, False, True), (5, 1024, True, True, True)]:
        qlt_case(nranks, ncells, pr, imb, partial, 0 if partial else 2)
    s2 = stats(dbl)
    assert s2["groups"] > s1["groups"] and s2["sends"] == s2["recvs"] and s2["sends"] > s1["sends"], (s1, s2)
    for nranks, ncells in [(2, 1000), (8, 5000)]:
        caas_case(nranks, ncells)
    s3 = stats(dbl)
    assert s3["allreduce"] > s2["allreduce"], (s2, s3)
    for nranks, ncell, tr, partial in [(2, 24, False, False), (3, 1000, True, False), (8, 1000, False, False),
                                       (8, 1000, True, True), (3, 1000, False, True)]:
        bfb_case(nranks, ncell, tr, partial)
    s4 = stats(dbl)
    assert s4["errors"] == 0, s4
    print(json.dumps(s4))


if __name__ == "__main__":
    main()
