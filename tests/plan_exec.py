"""Numpy executor of a product plan (tiles/stages/boundary exchange) -- TEST
INFRASTRUCTURE. It replays, on the CPU and in the same order as the HIP kernels,
what one rank does in QLT::run, using the oracle's node solver for the math. It
exists to test the HOST logic of the product (tiling, slot numbering, the
all-gather protocol, the replicated top of the tree) without a GPU, including
with world_size > 1 over gloo."""
import numpy as np

import oracle_lib as o

CONSERVE, SHAPEPRESERVE, CONSISTENT, NONNEGATIVE = 1, 2, 4, 8


def l2r_bulk(pt):
    if pt & NONNEGATIVE:
        return 2 if pt & CONSERVE else 1
    return 4 if pt & CONSERVE else 3


def field_ops(md, l2rndps):
    ops = np.zeros(l2rndps, np.int32)
    for ti, pt in enumerate(md["trcr2prob"]):
        if (pt & CONSISTENT) and not (pt & SHAPEPRESERVE):
            ops[md["trcr2bl2r"][ti]] = 1
            ops[md["trcr2bl2r"][ti] + 2] = 2
    return ops


def tile_local_nodes(stage, t):
    in0, nleaf, out, nint, koff = [int(x) for x in stage["tiles"][t][:5]]
    kids = stage["kids"][koff:koff + nint].astype(np.int64)
    return in0, nleaf, out, nint, kids


def up_stage(stage, data, ops, zero_init=False):
    """data[field][slot]; mirrors tree_up_kernel."""
    for t in range(len(stage["tiles"])):
        in0, nleaf, out, nint, kids = tile_local_nodes(stage, t)
        v = np.zeros((data.shape[0], nleaf + nint))
        v[:, :nleaf] = data[:, in0:in0 + nleaf]
        for j in range(nint):
            k0, k1 = v[:, kids[j, 0]], v[:, kids[j, 1]]
            if zero_init:
                s = (0.0 + k0) + k1
            else:
                s = k0 + k1
            mn = np.where(k0 < k1, k0, k1)  # comparisons, as cedr_kokkos.hpp:136-139
            mx = np.where(k0 > k1, k0, k1)
            v[:, nleaf + j] = np.where(ops == 0, s, np.where(ops == 1, mn, mx))
        data[:, out] = v[:, nleaf + nint - 1 if nint else 0]


def root_compute(md, root, l2r, r2l):
    for ti, pt in enumerate(md["trcr2prob"]):
        lb, rb = md["trcr2bl2r"][ti], md["trcr2br2l"][ti]
        os_ = l2r_bulk(pt) - 1 if pt & CONSERVE else (0 if pt & NONNEGATIVE else 1)
        r2l[rb, root] = l2r[lb + os_, root]
        if (pt & CONSISTENT) and not (pt & SHAPEPRESERVE):
            r2l[rb + 1, root] = l2r[lb, root]
            r2l[rb + 2, root] = l2r[lb + 2, root]


def down_stage(stage, md, l2r, r2l, prefer=False):
    """Mirrors qlt_down_kernel."""
    for t in range(len(stage["tiles"])):
        in0, nleaf, out, nint, kids = tile_local_nodes(stage, t)
        nn = nleaf + nint
        rho = np.zeros(nn)
        rho[:nleaf] = l2r[0, in0:in0 + nleaf]
        for j in range(nint):
            rho[nleaf + j] = rho[kids[j, 0]] + rho[kids[j, 1]]
        root = nn - 1 if nint else 0
        for ti, pt in enumerate(md["trcr2prob"]):
            pt = int(pt)
            lb, rb = md["trcr2bl2r"][ti], md["trcr2br2l"][ti]
            sp = bool(pt & SHAPEPRESERVE)
            t_only = bool(pt & CONSISTENT) and not sp
            rec = np.zeros((nn, 4))
            if sp:
                for f in range(3):
                    rec[:nleaf, f] = l2r[lb + f, in0:in0 + nleaf]
            else:
                rec[:nleaf, 1] = l2r[lb + (1 if pt & CONSISTENT else 0), in0:in0 + nleaf]
            for j in range(nint):
                rec[nleaf + j, :3] = rec[kids[j, 0], :3] + rec[kids[j, 1], :3]
            rec[root, 3] = r2l[rb, out]
            qlo = qhi = 0.0
            if t_only:
                qlo, qhi = r2l[rb + 1, out], r2l[rb + 2, out]
            for j in range(nint - 1, -1, -1):
                me, c0, c1 = nleaf + j, kids[j, 0], kids[j, 1]
                if t_only:
                    pd = [qlo, rec[me, 1], qhi]
                    k0d, k1d = [qlo, rec[c0, 1], qhi], [qlo, rec[c1, 1], qhi]
                elif pt & NONNEGATIVE:
                    pd = [0, 0, 0]
                    k0d, k1d = [rec[c0, 1], 0, 0], [rec[c1, 1], 0, 0]
                else:
                    pd, k0d, k1d = rec[me, :3], rec[c0, :3], rec[c1, :3]
                x = o.solve_node_problem(pt, rho[me], pd, rec[me, 3], rho[c0], k0d, rho[c1], k1d, prefer)
                rec[c0, 3], rec[c1, 3] = x
            r2l[rb, in0:in0 + nleaf] = rec[:nleaf, 3]
            if t_only:
                r2l[rb + 1, in0:in0 + nleaf] = qlo
                r2l[rb + 2, in0:in0 + nleaf] = qhi
                l2r[lb, in0:in0 + nleaf] = qlo
                l2r[lb + 2, in0:in0 + nleaf] = qhi


class RankState:
    """One rank's buffers and plan; the exchange is supplied by the caller."""

    def __init__(self, plan, md, prefer=False):
        self.plan, self.md, self.prefer = plan, md, prefer
        self.l2rndps, self.r2lndps = int(md["prob2bl2r"][6]), int(md["prob2br2l"][6])
        S = int(plan["stride"])
        self.l2r = np.zeros((self.l2rndps, S))
        self.r2l = np.zeros((self.r2lndps, S))
        self.ops = field_ops(md, self.l2rndps)

    def set(self, rhom, Qm, Qm_min, Qm_max, Qm_prev):
        n = len(rhom)
        self.l2r[0, :n] = rhom
        for ti, pt in enumerate(self.md["trcr2prob"]):
            lb = self.md["trcr2bl2r"][ti]
            if pt & SHAPEPRESERVE:
                self.l2r[lb, :n], self.l2r[lb + 1, :n], self.l2r[lb + 2, :n] = Qm_min[ti], Qm[ti], Qm_max[ti]
                nxt = 3
            elif pt & CONSISTENT:
                self.l2r[lb, :n], self.l2r[lb + 1, :n], self.l2r[lb + 2, :n] = (
                    Qm_min[ti] / rhom, Qm[ti], Qm_max[ti] / rhom)
                nxt = 3
            else:
                self.l2r[lb, :n] = Qm[ti]
                nxt = 1
            if pt & CONSERVE:
                self.l2r[lb + nxt, :n] = Qm_prev[ti]

    def up_local(self):
        for st in self.plan["local"]:
            up_stage(st, self.l2r, self.ops)

    def send_records(self):
        nb = int(self.plan["nb_max"])
        buf = np.zeros((nb, self.l2rndps))
        if nb == 0:  # single rank: nothing is exchanged
            return buf
        for j, s in enumerate(self.plan["my_root_slots"]):
            buf[j] = self.l2r[:, s]
        return buf

    def recv_records(self, allbuf):
        """allbuf[rank][j][field] from every rank (the all-gather result)."""
        gs = self.plan["gather_slots"]
        for r in range(gs.shape[0]):
            for j in range(gs.shape[1]):
                if gs[r, j] >= 0:
                    self.l2r[:, gs[r, j]] = allbuf[r][j]

    def top_and_down(self):
        for st in self.plan["top"]:
            up_stage(st, self.l2r, self.ops)
        root_compute(self.md, int(self.plan["root_slot"]), self.l2r, self.r2l)
        for st in reversed(self.plan["top"]):
            down_stage(st, self.md, self.l2r, self.r2l, self.prefer)
        if len(self.plan["top"]):
            for ts, ls in zip(self.plan["my_top_slots"], self.plan["my_root_slots"]):
                self.r2l[:, ls] = self.r2l[:, ts]
        for st in reversed(self.plan["local"]):
            down_stage(st, self.md, self.l2r, self.r2l, self.prefer)

    def get(self, n):
        return np.array([self.r2l[self.md["trcr2br2l"][ti], :n] for ti in range(len(self.md["trcr2prob"]))])
