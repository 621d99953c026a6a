"""The hand-derived vectors of tests/golden/hand_derived_vectors.{md,json} (worked out on paper
from reference cedr_local_inl.hpp:69-165 and cedr_qlt_inl.hpp:69-203, exact dyadic arithmetic)
against the oracle and against the product's host build of cedr::local."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
V = json.load(open(os.path.join(HERE, "golden", "hand_derived_vectors.json")))


@pytest.mark.parametrize("c", V["solve_1eq_bc_qp_2d"], ids=lambda c: c["name"])
def test_qp_2d(oracle, ca, c):
    info, x = oracle.solve_2d(c["w"], [1, 1], c["b"], c["xlo"], c["xhi"], c["y"], c["clip"], c["early"])
    assert info == c["info"] and list(x) == c["x"]
    info, x = ca.local_solve(ca.LOCAL_BC_QP_2D, c["w"], [1, 1], c["b"], c["xlo"], c["xhi"], c["y"],
                             clip=c["clip"], early_exit_on_tol=c["early"])
    assert info == c["info"] and list(x) == c["x"]


@pytest.mark.parametrize("c", V["solve_node_problem"], ids=lambda c: c["name"])
def test_node_problem(oracle, c):
    x = oracle.solve_node_problem(c["pt"], c["rhom"], c["pd"], c["Qm"], c["rhom0"], c["k0d"], c["rhom1"], c["k1d"])
    assert list(x) == c["x"]
