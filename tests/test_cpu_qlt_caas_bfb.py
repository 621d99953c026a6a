"""The reference's own QLT / CAAS / BfbTreeAllReducer unit tests restated against the
oracle, with the reference's tolerances, on 1, 3 and 8 simulated ranks."""
import numpy as np
import pytest

import common
import randomized as R

EPS = np.finfo(float).eps


# unittest_QLT, cedr_qlt.cpp:745-772
@pytest.mark.parametrize("P", [1, 3, 8])
@pytest.mark.parametrize("mult", [1, 2, 7, 21])
@pytest.mark.parametrize("pseudorandom", [False, True])
@pytest.mark.parametrize("imbalanced", [False, True])
def test_unittest_qlt(oracle, P, mult, pseudorandom, imbalanced):
    nc = P * mult
    trs = R.tracers_vector()
    for prefer in (False, True):
        v = R.make_values(trs, nc, seed=17 * nc + P)
        out = common.run_oracle_qlt(oracle, (pseudorandom, imbalanced), nc, P,
                                    [t.problem_type for t in trs], v, prefer)
        nerr, msgs = R.check(trs, v, out, prefer, verbose=False)
        assert nerr == 0, msgs[:3]


def test_qlt_rank_invariance(oracle):
    """SURVEY.md 3.2 [probe]: get_Qm bit patterns are identical across rank counts,
    ownership patterns -- one set of outputs pins every decomposition."""
    trs = R.tracers_vector()
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, 333, seed=5)
    base = common.run_oracle_qlt(oracle, (False, False), 333, 1, types, v)
    for P, pr in ((3, False), (8, False), (8, True)):
        assert np.array_equal(common.run_oracle_qlt(oracle, (pr, False), 333, P, types, v), base)


# caas::test::unittest, cedr_caas.cpp:381-393
@pytest.mark.parametrize("P", [1, 3, 8])
@pytest.mark.parametrize("nlcl", [1, 2, 4, 11])
@pytest.mark.parametrize("own_reducer", [False, True])
def test_unittest_caas(oracle, P, nlcl, own_reducer):
    ncells = P * nlcl
    if ncells > P:
        ncells -= P // 2
    per = [ncells // P + (1 if r < ncells - (ncells // P) * P else 0) for r in range(P)]
    trs = R.caas_tracers(R.tracers_vector())
    v = R.make_values(trs, ncells, seed=ncells + P)
    accum = 0
    if own_reducer:  # one n_accum_in_place for all simulated ranks
        accum = 3 if all(n % 3 == 0 for n in per) else 2 if all(n % 2 == 0 for n in per) else 1
    out, _ = common.run_oracle_caas(oracle, per, [t.problem_type for t in trs], v, accum)
    nerr, msgs = R.check(trs, v, out, verbose=False)
    assert nerr == 0, msgs[:3]


# BfbTreeAllReducer::unittest, cedr_bfb_tree_allreduce.cpp:162-220
@pytest.mark.parametrize("P", [1, 3, 8])
@pytest.mark.parametrize("mult", [1, 3])
@pytest.mark.parametrize("pseudorandom", [False, True])
@pytest.mark.parametrize("imbalanced", [False, True])
@pytest.mark.parametrize("transpose", [False, True])
def test_unittest_bfb(oracle, P, mult, pseudorandom, imbalanced, transpose):
    ncell, nfield = P * mult, 3
    rng = np.random.default_rng(ncell)
    tree = oracle.Tree.make_1d(ncell, P, pseudorandom, imbalanced)
    sends, plain = [], np.zeros(nfield)
    for r in range(P):
        nl = len(oracle.analyze(tree, ncell, r)["levels"][0]["nodes"])
        s = rng.random(nl * nfield)
        sends.append(s)
        plain += s.reshape(nfield, nl).sum(1) if transpose else s.reshape(nl, nfield).sum(0)
    recv = oracle.bfb_allreduce(tree, ncell, P, nfield, transpose, sends)
    for r in range(1, P):
        assert np.array_equal(recv[r], recv[0])
    re = np.abs(recv[0] - plain) / np.maximum(np.abs(recv[0]), np.abs(plain))
    assert re.max() <= max(2 * np.log(max(ncell, 2)), 4) * EPS


def test_bfb_decomposition_invariance(oracle):
    """Same tree, same leaf values, different ownership: bit-for-bit the same sums."""
    ncell, nfield = 96, 2
    rng = np.random.default_rng(9)
    vals = rng.random((ncell, nfield))
    res = []
    for P, pr in ((1, False), (4, False), (4, True), (8, True)):
        tree = oracle.Tree.make_1d(ncell, P, pr, False)
        sends = []
        for r in range(P):
            ns = oracle.analyze(tree, ncell, r)
            gids = [int(ns["nodes"][i][1]) for i in ns["levels"][0]["nodes"]]
            sends.append(vals[gids].ravel())
        res.append(oracle.bfb_allreduce(tree, ncell, P, nfield, False, sends)[0])
    for x in res[1:]:
        assert np.array_equal(x, res[0])
