"""Caller trees other than the 1-D mesh bisection, with ownership -- TEST INFRASTRUCTURE.

random_tree: binary tree over nleaf leaves as arrays (what cedr_tree_create takes): 'merge' =
repeatedly merge two adjacent subtrees picked at random (bushy, irregular), 'chain' = left-deep
caterpillar (depth nleaf - 1: every stage of the plan sees a degenerate tile), 'skew' = merges
biased towards one end. Global cell ids are a random permutation of the leaves.

assign_ranks: who owns what, for the multi-rank tests: 'blocks' = runs of consecutive leaves per
rank (what a space-filling curve gives), 'scattered' = every leaf drawn at random; pin > 0 also
pins that share of the internal nodes to random ranks (a rank may then own a node without owning
anything below it; allowed by tree::Node::rank, cedr_tree_caller.hpp:15)."""
import numpy as np


def random_tree(rng, nleaf, kind):
    kid0, kid1 = [-1] * nleaf, [-1] * nleaf
    roots = list(range(nleaf))
    while len(roots) > 1:
        if kind == "chain":
            i = 0
        elif kind == "skew":
            i = int(min(len(roots) - 2, rng.integers(0, max(1, len(roots) // 8))))
        else:
            i = int(rng.integers(0, len(roots) - 1))
        kid0.append(roots[i]); kid1.append(roots[i + 1])
        roots[i:i + 2] = [len(kid0) - 1]
    n = len(kid0)
    rank = np.full(n, -1, np.int32)
    rank[:nleaf] = 0
    cell = np.full(n, -1, np.int64)
    cell[:nleaf] = rng.permutation(nleaf)
    return roots[0], kid0, kid1, rank, cell


def assign_ranks(rng, nleaf, rank, nranks, how="blocks", pin=0.0):
    """Returns a copy of `rank` (leaves are entries 0..nleaf-1, in left-to-right order) with every
    rank owning at least one leaf."""
    rank = np.array(rank, copy=True)
    if how == "blocks":
        cuts = np.sort(rng.choice(np.arange(1, nleaf), size=nranks - 1, replace=False)) if nranks > 1 else []
        bounds = [0] + [int(c) for c in cuts] + [nleaf]
        owners = rng.permutation(nranks)
        for r in range(nranks):
            rank[bounds[r]:bounds[r + 1]] = owners[r]
    else:
        rank[:nleaf] = rng.integers(0, nranks, nleaf)
        rank[rng.choice(nleaf, size=nranks, replace=False)] = np.arange(nranks)
    if pin > 0:
        internal = np.arange(nleaf, len(rank))
        chosen = internal[rng.random(len(internal)) < pin]
        rank[chosen] = rng.integers(0, nranks, len(chosen))
    return rank


def oracle_single_rank(oracle, root, kid0, kid1, cell, nleaf, types, v, prefer=False):
    """The oracle's QLT on the same tree with one rank: what any rank count and ownership must give."""
    n = len(kid0)
    r1 = np.full(n, -1, np.int32)
    r1[:nleaf] = 0
    oq = oracle.OracleQLT(oracle.Tree.from_arrays(root, kid0, kid1, r1, cell), nleaf, 1, prefer)
    oq.declare(types)
    g = oq.owned_glblcells(0)
    oq.set(0, v.rhom[g], v.Qm[:, g], v.Qm_min[:, g], v.Qm_max[:, g], v.Qm_prev[:, g])
    oq.run()
    want = np.zeros_like(v.Qm)
    want[:, g] = oq.get(0)
    return want
