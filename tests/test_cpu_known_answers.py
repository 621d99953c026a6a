"""Oracle and product host code against the reference's known answers recorded in
SURVEY.md (tests/golden/survey_known_answers.json). No GPU needed."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "survey_known_answers.json")))


def _analyzers(oracle, ca):
    return [("oracle", lambda nc, P, r: oracle.analyze(oracle.Tree.make_1d(nc, P), nc, r)),
            ("product", lambda nc, P, r: ca.Tree.make_1d(nc, P).analyze(nc, r))]


def test_six_cells_one_rank(oracle, ca):
    ka = KA["six_cells_one_rank"]
    for name, an in _analyzers(oracle, ca):
        ns = an(6, 1, 0)
        assert ns["nslots"] == ka["nslots"], name
        assert [len(l["nodes"]) for l in ns["levels"]] == ka["level_sizes"], name
        off = lambda idx: int(ns["nodes"][idx][6])
        kids = lambda idx: [off(k) for k in ns["nodes"][idx][4:6]]
        assert sorted(off(i) for i in ns["levels"][0]["nodes"]) == ka["leaf_offsets"]
        for lvl, key in ((1, "level1"), (2, "level2")):
            got = [{"offset": off(i), "kid_offsets": kids(i)} for i in ns["levels"][lvl]["nodes"]]
            assert got == ka[key], (name, key)
        assert off(ns["levels"][3]["nodes"][0]) == ka["root_offset"]


def test_six_cells_two_ranks(oracle, ca):
    ka = KA["six_cells_two_ranks"]
    for name, an in _analyzers(oracle, ca):
        r0, r1 = an(6, 2, 0), an(6, 2, 1)
        assert r0["nslots"] == ka["rank0"]["nslots"] and r1["nslots"] == ka["rank1"]["nslots"], name
        assert r0["levels"][ka["rank0"]["root_level"]]["kids"].tolist() == ka["rank0"]["kids_msgs_at_root_level"]
        assert r1["levels"][2]["me"].tolist() == ka["rank1"]["me_msgs_at_level2"]


@pytest.mark.parametrize("nc", [1024, 1000])
def test_level_sizes(oracle, ca, nc):
    for name, an in _analyzers(oracle, ca):
        ns = an(nc, 1, 0)
        assert [len(l["nodes"]) for l in ns["levels"]] == KA["level_sizes"][str(nc)], name


def test_buffer_sizes_config_A(oracle, ca):
    q = oracle.OracleQLT(oracle.Tree.make_1d(1024), 1024)
    q.declare([7] * 4)
    l2r, r2l = q.ndps()
    assert [q.nslots() * l2r, q.nslots() * r2l] == KA["qlt_buffer_sizes"]["A_1024x4_cst"]
    # product: host-side setup needs no GPU (device work starts at finish_setup)
    pq = ca.QLT(ca.Parallel.self_(), 1024, ca.Tree.make_1d(1024))
    for _ in range(4):
        pq.declare_tracer(7, 0)
    pq.end_tracer_declarations()
    assert list(pq.get_buffers_sizes()) == KA["qlt_buffer_sizes"]["A_1024x4_cst"]
    c = ca.CAAS(ca.Parallel.self_(), 1024)
    for _ in range(4):
        c.declare_tracer(7, 0)
    c.end_tracer_declarations()
    assert list(c.get_buffers_sizes()) == KA["caas_buffer_sizes"]["A_1024x4_cst"]


def test_buffer_size_formulas_config_C():
    """C (1.5M x 40 cst): nslots = 2n-1, l2rndps = 1+4nt, r2lndps = nt; CAAS per
    cedr_caas.cpp:76-91. Checked as arithmetic; building the 1.5M tree is left to the GPU bench."""
    n, nt = 1500000, 40
    assert [(2 * n - 1) * (1 + 4 * nt), (2 * n - 1) * nt] == KA["qlt_buffer_sizes"]["C_1500000x40_cst"]
    assert [n * (4 * nt + 1), 8 * nt] == KA["caas_buffer_sizes"]["C_1500000x40_cst"]
