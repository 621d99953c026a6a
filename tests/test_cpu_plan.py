"""Host logic of the product that decides WHAT the GPU does: the tile/stage plan and
the one-all-gather multi-rank protocol. A numpy executor (plan_exec.py, test
infrastructure) replays the plan on the CPU with the oracle's node solver; results
must be bit-identical to the oracle's single-rank QLT whatever the tile size, the
rank count or the ownership pattern. The world_size-2 case also runs over real
torch.distributed (gloo) processes."""
import os
import subprocess
import sys

import numpy as np
import pytest

import common
import plan_exec
import randomized as R

HERE = os.path.dirname(os.path.abspath(__file__))


def planning_parallel(ca, rank, size):
    if size == 1:
        return ca.Parallel.self_()
    bad = lambda *a: 1  # planning never communicates
    # (with the third callback: a transport without point-to-point exchange plans every full tree for
    # the all-gather and refuses partial trees)
    return ca.Parallel.callbacks_p2p(rank, size, bad, bad, bad)


def make_rank(ca, nc, P, r, tree_args, types, tile=None, monkeypatch=None):
    tree = ca.Tree.make_1d(nc, P, *tree_args)
    q = ca.QLT(planning_parallel(ca, r, P), nc, tree)
    for t in types:
        q.declare_tracer(t, 0)
    q.end_tracer_declarations()
    return q, plan_exec.RankState(q.plan(), q.metadata())


def run_plan(ca, nc, P, tree_args, types, v, prefer=False, use_apex=False):
    ranks = [make_rank(ca, nc, P, r, tree_args, types) for r in range(P)]
    for q, st in ranks:
        st.prefer = prefer
        g = q.get_owned_glblcells()
        st.set(v.rhom[g], v.Qm[:, g], v.Qm_min[:, g], v.Qm_max[:, g], v.Qm_prev[:, g])
        st.up_local()
    allbuf = [st.send_records() for _, st in ranks]  # the all-gather
    out = np.zeros_like(v.Qm)
    for q, st in ranks:
        if P > 1:
            st.recv_records(allbuf)
        if use_apex:
            assert st.plan["apex"] is not None
        st.top_and_down(use_apex)
        g = q.get_owned_glblcells()
        out[:, g] = st.get(len(g))
    return out


@pytest.mark.parametrize("micro", ["0", "1", "cut0", "cut1"])
@pytest.mark.parametrize("nc,P,pseudorandom,imbalanced", [
    (1, 1, False, False), (2, 1, False, False), (37, 1, False, True), (200, 1, False, False),
    (700, 1, False, False), (2, 2, False, False), (24, 2, False, False), (60, 3, False, True),
    (64, 8, False, False), (56, 8, True, False), (40, 4, True, True)])
def test_plan_execution_matches_oracle(ca, oracle, nc, P, pseudorandom, imbalanced, micro, monkeypatch):
    monkeypatch.setenv("CEDR_HIP_TILE_LEAVES", "8")  # several stages even on small trees
    monkeypatch.setenv("CEDR_HIP_MICRO", "0" if micro == "0" else "1")
    if micro.startswith("cut"):
        monkeypatch.setenv("CEDR_HIP_MICRO_CUT", micro[3:])
    trs = R.tracers_vector()[:12]  # all six problem types, two perturbations
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, nc, seed=nc + P)
    want = common.run_oracle_qlt(oracle, (False, imbalanced), nc, 1, types, v)
    got = run_plan(ca, nc, P, (pseudorandom, imbalanced), types, v)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("nc,P,pseudorandom,imbalanced", [
    (2, 2, False, False), (24, 2, False, False), (60, 3, False, True), (64, 8, False, False),
    (56, 8, True, False), (40, 4, True, True), (300, 7, False, True), (640, 64, False, False)])
def test_apex_tile_matches_oracle(ca, oracle, nc, P, pseudorandom, imbalanced, monkeypatch):
    """The multi-rank top as the one plain tile the apex kernels sweep (plan["apex"], <= 64
    boundary roots): gathered records in, root problem at the tile's output, masses back at the
    ranks' top slots -- the same bits as the staged top and as the oracle's single-rank tree."""
    monkeypatch.setenv("CEDR_HIP_TILE_LEAVES", "8")
    trs = R.tracers_vector()[:12]
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, nc, seed=3*nc + P)
    want = common.run_oracle_qlt(oracle, (False, imbalanced), nc, 1, types, v)
    got = run_plan(ca, nc, P, (pseudorandom, imbalanced), types, v, use_apex=True)
    assert np.array_equal(got, want)


def test_apex_tile_shape(ca):
    """One tile, plain (no micro table), one output; its inputs are exactly the gathered slots;
    more than 64 boundary roots: no apex tile, the staged top is used."""
    nc, P = 512, 8
    tree = ca.Tree.make_1d(nc, P, True, False)  # pseudorandom ownership: many boundary roots
    for r in range(P):
        pl = tree.plan(nc, r, P, 0)
        gs = pl["gather_slots"]
        nb = int((gs >= 0).sum())
        ap = pl["apex"]
        if nb > 64:
            assert ap is None
            continue
        assert len(ap["tiles"]) == 1 and len(ap["micro"]) == 0
        in0, nleaf, out, nint, koff, loff, nlvl, moff, nout, ooff = [int(x) for x in ap["tiles"][0]]
        assert nleaf == nb and nint == nb - 1 and nout == 1 and out < pl["nslots"]
        assert set(range(in0, in0 + nleaf)) == set(int(s) for s in gs.ravel() if s >= 0)
    pl = ca.Tree.make_1d(1024, 128).plan(1024, 5, 128, 0)
    assert pl["apex"] is None and len(pl["top"]) >= 1
    assert ca.Tree.make_1d(64, 1).plan(64, 0, 1, 0)["apex"] is None  # single rank: no top at all


def run_level_plan(ca, nc, P, full_tree, types, v, partial):
    """Level mode with P rank threads and an in-process mailbox; every rank plans from its own
    (pruned, if partial) tree."""
    import threading
    box = plan_exec.Mailbox()
    out = np.zeros_like(v.Qm)
    errs = []

    def rank_main(r):
        try:
            tree = full_tree.prune(r) if partial else full_tree
            q = ca.QLT(planning_parallel(ca, r, P), nc, tree)
            for t in types:
                q.declare_tracer(t, 0)
            q.end_tracer_declarations()
            pl = q.plan()
            assert pl["level_mode"] and not pl["top"] and pl["apex"] is None
            st = plan_exec.RankState(pl, q.metadata())
            st.owns_root = pl["root_owner"] == r
            g = q.get_owned_glblcells()
            st.set(v.rhom[g], v.Qm[:, g], v.Qm_min[:, g], v.Qm_max[:, g], v.Qm_prev[:, g])
            st.up_local()

            def xch(tag, sends, recvs):
                for peer, a in sends:
                    box.put((r, peer, tag), a)
                return [box.get((peer, r, tag)) for peer, _ in recvs]
            st.run_level_mode(xch)
            out[:, g] = st.get(len(g))
        except Exception as e:  # noqa: BLE001
            import traceback
            errs.append("rank %d: %s" % (r, traceback.format_exc()))
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, "\n".join(errs)
    return out


@pytest.mark.parametrize("partial", [False, True])
@pytest.mark.parametrize("nc,P,pseudorandom,imbalanced", [
    (2, 2, False, False), (24, 2, False, False), (60, 3, False, True), (64, 8, False, False),
    (56, 8, True, False), (40, 4, True, True), (300, 7, False, True), (200, 5, True, False)])
def test_level_mode_matches_oracle(ca, oracle, nc, P, pseudorandom, imbalanced, partial, monkeypatch):
    """Partial caller trees (tree::Node::level) and scattered ownership: pure subtrees by tiles, the
    rest of a rank's nodes level by level with point-to-point messages -- the same bits as the
    oracle's single-rank tree."""
    monkeypatch.setenv("CEDR_HIP_TILE_LEAVES", "8")
    if not partial:
        monkeypatch.setenv("CEDR_HIP_EXCHANGE", "2")  # a full tree: level mode has to be asked for
    trs = R.tracers_vector()[:12]
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, nc, seed=5*nc + P)
    want = common.run_oracle_qlt(oracle, (False, imbalanced), nc, 1, types, v)
    got = run_level_plan(ca, nc, P, ca.Tree.make_1d(nc, P, pseudorandom, imbalanced), types, v, partial)
    assert np.array_equal(got, want)


def run_tree_plan(ca, nc, P, tree, types, v, mode, partial):
    """One tree on P ranks through the numpy plan executor: mode 1 = all-gather + replicated top,
    2 = level mode, 0 = whichever the planner picks; partial: every rank plans from its pruned tree (then always level mode)."""
    import threading
    box = plan_exec.Mailbox()
    out = np.zeros_like(v.Qm)
    states, errs = [None] * P, []
    for r in range(P):
        q = ca.QLT(planning_parallel(ca, r, P), nc, tree.prune(r) if partial else tree)
        for t in types:
            q.declare_tracer(t, 0)
        q.end_tracer_declarations()
        pl = q.plan()
        if mode == 0:  # the planner's own choice; every rank must make the same one
            chosen = pl["level_mode"] if r == 0 else chosen
            assert pl["level_mode"] == chosen
        else:
            assert pl["level_mode"] == (mode == 2 or partial)
        st = plan_exec.RankState(pl, q.metadata())
        st.owns_root = pl["root_owner"] == r
        g = q.get_owned_glblcells()
        st.set(v.rhom[g], v.Qm[:, g], v.Qm_min[:, g], v.Qm_max[:, g], v.Qm_prev[:, g])
        st.up_local()
        states[r] = (q, st, g)
    if mode == 0:
        mode = 2 if chosen else 1
    if mode == 1 and not partial:
        allbuf = [st.send_records() for _, st, _ in states]
        for q, st, g in states:
            st.recv_records(allbuf)
            st.top_and_down(st.plan["apex"] is not None)
            out[:, g] = st.get(len(g))
        return out

    def rank_main(r):
        try:
            q, st, g = states[r]

            def xch(tag, sends, recvs):
                for peer, a in sends:
                    box.put((r, peer, tag), a)
                return [box.get((peer, r, tag)) for peer, _ in recvs]
            st.run_level_mode(xch)
            out[:, g] = st.get(len(g))
        except Exception:  # noqa: BLE001
            import traceback
            errs.append("rank %d: %s" % (r, traceback.format_exc()))
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, "\n".join(errs)
    return out


@pytest.mark.parametrize("kind,nleaf,P,how,pin", [
    ("merge", 40, 2, "blocks", 0.0), ("merge", 333, 5, "blocks", 0.0), ("merge", 200, 4, "scattered", 0.0),
    ("skew", 150, 3, "blocks", 0.0), ("chain", 60, 3, "blocks", 0.0), ("chain", 45, 4, "scattered", 0.0),
    ("merge", 120, 4, "blocks", 0.3), ("skew", 90, 6, "scattered", 0.5)])
def test_random_caller_trees_on_many_ranks(ca, oracle, kind, nleaf, P, how, pin, monkeypatch):
    """Irregular caller trees (cedr_tree_caller.hpp:12-30) with block and scattered ownership, and
    internal nodes pinned to ranks that own nothing below them: the all-gather path, level mode on
    the full tree and level mode from per-rank partial trees all give the single-rank oracle's bits."""
    import random_trees as RT
    monkeypatch.setenv("CEDR_HIP_TILE_LEAVES", "8")
    rng = np.random.default_rng(7 * nleaf + P)
    root, kid0, kid1, rank, cell = RT.random_tree(rng, nleaf, kind)
    rank = RT.assign_ranks(rng, nleaf, rank, P, how, pin)
    trs = R.tracers_vector()[:12]
    types = [t.problem_type for t in trs]
    v = R.make_values(trs, nleaf, seed=nleaf)
    want = RT.oracle_single_rank(oracle, root, kid0, kid1, cell, nleaf, types, v)
    tree = ca.Tree.from_arrays(root, kid0, kid1, rank, cell)
    # mode 0 = the planner's own choice: nodes pinned away from the rank that owns everything below
    # them (pin > 0) make it pick level mode; the all-gather path takes ownership from the leaves
    for mode, partial in ((0 if pin > 0 else 1, False), (2, False), (2, True)):
        monkeypatch.setenv("CEDR_HIP_EXCHANGE", str(mode))
        if partial:
            monkeypatch.delenv("CEDR_HIP_EXCHANGE")
        got = run_tree_plan(ca, nleaf, P, tree, types, v, mode, partial)
        assert np.array_equal(got, want), (mode, partial)


def test_level_mode_is_chosen_for_scattered_ownership(ca):
    """The automatic rule: contiguous ownership keeps the one all-gather, pseudorandom ownership of
    a tree big enough for it to matter goes level by level; every rank decides alike."""
    nc, P = 100000, 4
    for pr, want in ((False, False), (True, True)):
        tree = ca.Tree.make_1d(nc, P, pr, False)
        modes = {tree.plan(nc, r, P, 0)["level_mode"] for r in range(P)}
        assert modes == {want}
    pl = ca.Tree.make_1d(nc, P, True, False).prune(1).plan(nc, 1, P, 0)
    assert pl["level_mode"]
    # a pruned tree is O(cells/ranks + depth)
    assert ca.Tree.make_1d(nc, P).prune(2).nnodes() < 2*nc//P + 64


@pytest.mark.parametrize("tile", [2, 3, 5, 64, 256])
def test_plan_invariants(ca, tile):
    nc = 1000
    tree = ca.Tree.make_1d(nc, 1, False, True)
    pl = tree.plan(nc, 0, 1, -tile)  # micro tiles off: every tile input is one slot
    assert pl["nlcl"] == nc
    seen = np.zeros(pl["nslots"], np.int32)
    covered = np.zeros(pl["nslots"], np.int32)
    for si, st in enumerate(pl["local"]):
        # the fused top tile: one plain tile of up to 2048 inputs, only ever the last stage
        assert not st["top"] or (si == len(pl["local"]) - 1 and si > 0 and len(st["tiles"]) == 1)
        for t in st["tiles"]:
            in0, nleaf, out, nint, koff, loff, nlvl = [int(x) for x in t[:7]]
            assert 1 <= nleaf <= (2048 if st["top"] else tile) and nint == nleaf - 1
            assert int(t[8]) == 1 and st["outs"][int(t[9])] == nleaf + nint - 1  # one output: the root
            covered[in0:in0 + nleaf] += 1
            seen[out] += 1
            kids = st["kids"][koff:koff + nint]
            for j in range(nint):  # kids come before parents
                assert (kids[j] < nleaf + j).all()
            used = np.bincount(kids.ravel(), minlength=nleaf + nint)
            assert (used[:nleaf + nint - 1] == 1).all()  # a tree: every non-root node used once
            lp = st["lvlptr"][loff:loff + nlvl + 1]
            assert lp[0] == 0 and lp[-1] == nint and (np.diff(lp) > 0).all()
    assert (seen[nc:] == 1).all() and (seen[:nc] == 0).all()
    assert (covered[:pl["root_slot"]] == 1).all() and covered[pl["root_slot"]] == 0
    assert pl["stride"] % 16 == 0 and pl["stride"] >= pl["nslots"]


@pytest.mark.parametrize("imbalanced", [False, True])
def test_micro_plan_invariants(ca, imbalanced):
    """Every stage uses micro tiles: <= 64 micro-subtrees of <= 3 consecutive inputs per
    tile; the first stage's inputs are the cells, a later stage's inputs are exactly the
    previous stage's output slots; every input is in exactly one micro-subtree."""
    nc = 5000
    pl = ca.Tree.make_1d(nc, 1, False, imbalanced).plan(nc, 0, 1, 0)
    want = set(range(nc))  # slots the stage must consume
    assert len(pl["local"]) >= 2
    assert pl["local"][-1]["top"]  # 5000 cells: one micro stage, then the fused top tile
    for si, st in enumerate(pl["local"]):
        if st["top"]:
            # The fused top tile: ONE plain tile whose inputs are exactly the previous stage's outputs
            # (consecutive slots) and whose one output is the root.
            assert si == len(pl["local"]) - 1 and len(st["tiles"]) == 1 and len(st["micro"]) == 0
            in0, nleaf, out, nint, koff, loff, nlvl, moff, nout, ooff = [int(x) for x in st["tiles"][0]]
            assert nleaf <= 2048 and nint == nleaf - nout and nout == 1
            assert set(range(in0, in0 + nleaf)) == want
            want = set(range(out, out + nout))
            continue
        assert len(st["micro"]) > 0
        got, produced = [], set()
        for t in st["tiles"]:
            in0, nleaf, out, nint, koff, loff, nlvl, moff, nout, ooff = [int(x) for x in t]
            assert 1 <= nleaf <= 64 and nint == nleaf - nout
            assert nlvl <= 4  # with a fused top every micro stage cuts its tiles (the first to 3 levels, later ones to 4)
            outs = st["outs"][ooff:ooff + nout]
            assert len(set(outs)) == nout and (outs < nleaf + nint).all()
            prev_end = 0
            for d in st["micro"][moff:moff + nleaf]:
                start, shape = int(d & 0xff), int(d >> 8)
                size = 1 if shape == 0 else 2 if shape == 1 else 3
                assert start == prev_end and 0 <= shape <= 3
                got.extend(range(in0 + start, in0 + start + size))
                prev_end = start + size
            produced.update(range(out, out + nout))
        assert len(got) == len(set(got)) and set(got) == want
        want = produced
    assert want == {pl["root_slot"]}


def test_multirank_plan_shapes(ca):
    nc, P = 64, 8
    tree = ca.Tree.make_1d(nc, P)
    for r in range(P):
        pl = tree.plan(nc, r, P, 128)
        assert pl["nlcl"] == 8 and pl["nb_max"] == 1 and pl["nmyroots"] == 1
        gs = pl["gather_slots"]
        assert gs.shape == (P, 1) and len(set(gs.ravel())) == P
        assert pl["my_top_slots"][0] == gs[r, 0]
        assert len(pl["top"]) == 1 and len(pl["top"][0]["tiles"]) == 1


def test_gloo_two_ranks():
    """world_size 2 over real processes: torch.distributed (gloo) all_gather carries
    the subtree-root records between the two ranks' plan executors."""
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.dirname(HERE), HERE, env.get("PYTHONPATH", "")])
    res = subprocess.run([sys.executable, os.path.join(HERE, "mp_gloo_worker.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "GLOO_OK" in res.stdout


def test_plan_invariants_under_sanitizers():
    """tests/cpp/plan_check.cpp: analyze + plan for ~7000 (tree, ownership, option) cases,
    built with -fsanitize=address,undefined, checking every index range the kernels rely on."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    subprocess.check_call(["make", "-s", "-C", d, "plan_check_asan"])
    res = subprocess.run([os.path.join(d, "plan_check_asan")], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "PASS" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


def test_transport_without_point_to_point_refuses_what_it_would_drop(ca, capfd, monkeypatch):
    """mpi::Comm::has_exchange() false (two-callback transport): a tree::Node::rank pin that only level mode
    honours is an error at construction (it used to be dropped silently); scattered ownership of a full tree
    still plans for the all-gather, with a warning on rank 0's stderr."""
    bad = lambda *a: 1
    # four cells, ranks 0 0 1 1; the node over cells 0 and 1 pinned to rank 1
    kid0 = [-1, -1, -1, -1, 0, 2, 4]
    kid1 = [-1, -1, -1, -1, 1, 3, 5]
    cell = [0, 1, 2, 3, -1, -1, -1]
    for pinned in (False, True):
        rank = [0, 0, 1, 1, 1 if pinned else -1, -1, -1]
        tree = lambda: ca.Tree.from_arrays(6, kid0, kid1, rank, cell)
        ca.QLT(ca.Parallel.callbacks_p2p(0, 2, bad, bad, bad), 4, tree())  # with messages: fine either way
        if pinned:
            with pytest.raises(ca.CedrError, match="pins a node"):
                ca.QLT(ca.Parallel.callbacks(0, 2, bad, bad), 4, tree())
            # ... and so is insisting on the all-gather design (exchange = 1), which would compute the node where
            # its cells are: refused, with or without point-to-point messages
            monkeypatch.setenv("CEDR_HIP_EXCHANGE", "1")
            for par in (ca.Parallel.callbacks(0, 2, bad, bad), ca.Parallel.callbacks_p2p(0, 2, bad, bad, bad)):
                with pytest.raises(ca.CedrError, match="pins a node"):
                    ca.QLT(par, 4, tree())
            monkeypatch.delenv("CEDR_HIP_EXCHANGE")
        else:
            ca.QLT(ca.Parallel.callbacks(0, 2, bad, bad), 4, tree())
    nc = 40000  # pseudorandom ownership: level mode by the planner's own rule
    capfd.readouterr()
    q = ca.QLT(ca.Parallel.callbacks(0, 4, bad, bad), nc, ca.Tree.make_1d(nc, 4, True, False))
    q.declare_tracer(7, 0)
    q.end_tracer_declarations()
    assert not q.plan()["level_mode"]
    assert "falling back to one all-gather" in capfd.readouterr().err


def test_cuts_follow_the_tracer_list(ca, monkeypatch):
    """The plan's cuts are chosen when the tracer list is known (QLT::end_tracer_declarations): up to 64 tracers, all shape
    preserving: first-stage tiles keep three LDS levels, later ones four; many tracers or any other problem type: two and
    three (round 4's). The environment's CEDR_HIP_MICRO_CUT / CUT2 override the rule."""
    for k in ("CEDR_HIP_MICRO_CUT", "CEDR_HIP_MICRO_CUT2", "CEDR_HIP_MICRO", "CEDR_HIP_TOP", "CEDR_HIP_TILE_LEAVES"):
        monkeypatch.delenv(k, raising=False)
    n = 200000

    def first_stage_levels(types):
        q = ca.QLT(ca.Parallel.self_(), n, ca.Tree.make_1d(n))
        for t in types:
            q.declare_tracer(t, 0)
        q.end_tracer_declarations()
        st = q.plan()["local"]
        return max(int(t[6]) for t in st[0]["tiles"]), [len(s["tiles"]) for s in st]
    assert first_stage_levels([7] * 40)[0] == 3
    assert first_stage_levels([7, 3, 2] * 10)[0] == 3       # cst, cs, s: all shape preserving
    assert first_stage_levels([7] * 65)[0] == 2             # many tracers
    assert first_stage_levels([7] * 39 + [4])[0] == 2       # a t-only tracer among them
    assert first_stage_levels([9])[0] == 2                  # nonnegative
    monkeypatch.setenv("CEDR_HIP_MICRO_CUT", "3")
    assert first_stage_levels([7] * 65)[0] == 3             # the environment decides
