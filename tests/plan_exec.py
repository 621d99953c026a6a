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


def tile_outs(stage, t):
    """Local node ids of the tile's outputs; output k lives in slot out + k."""
    nout, ooff = int(stage["tiles"][t][8]), int(stage["tiles"][t][9])
    return [int(x) for x in stage["outs"][ooff:ooff + nout]]


def tile_micro(stage, t):
    """Micro tiles: list of (first cell relative to in0, shape) per LDS-tree input, or None."""
    if not len(stage.get("micro", ())):
        return None
    nleaf, moff = int(stage["tiles"][t][1]), int(stage["tiles"][t][7])
    d = stage["micro"][moff:moff + nleaf].astype(np.int64)
    return [(int(x & 0xff), int(x >> 8)) for x in d]


def _comb(ops, k0, k1, zero_init):
    s = (0.0 + k0) + k1 if zero_init else k0 + k1
    mn = np.where(k0 < k1, k0, k1)  # comparisons, as cedr_kokkos.hpp:136-139
    mx = np.where(k0 > k1, k0, k1)
    return np.where(ops == 0, s, np.where(ops == 1, mn, mx))


def micro_cells(start, shape):
    return [start + k for k in range(1 if shape == 0 else 2 if shape == 1 else 3)]


def up_stage(stage, data, ops, zero_init=False):
    """data[field][slot]; mirrors tree_up_kernel."""
    for t in range(len(stage["tiles"])):
        in0, nleaf, out, nint, kids = tile_local_nodes(stage, t)
        v = np.zeros((data.shape[0], nleaf + nint))
        mic = tile_micro(stage, t)
        if mic is None:
            v[:, :nleaf] = data[:, in0:in0 + nleaf]
        else:
            for i, (start, shape) in enumerate(mic):
                c = [data[:, in0 + k] for k in micro_cells(start, shape)]
                if shape == 0:
                    v[:, i] = c[0]
                elif shape == 1:
                    v[:, i] = _comb(ops, c[0], c[1], zero_init)
                elif shape == 2:
                    v[:, i] = _comb(ops, _comb(ops, c[0], c[1], zero_init), c[2], zero_init)
                else:
                    v[:, i] = _comb(ops, c[0], _comb(ops, c[1], c[2], zero_init), zero_init)
        for j in range(nint):
            v[:, nleaf + j] = _comb(ops, v[:, kids[j, 0]], v[:, kids[j, 1]], zero_init)
        for k, lid in enumerate(tile_outs(stage, t)):
            data[:, out + k] = v[:, lid]


def root_compute(md, root, l2r, r2l):
    for ti, pt in enumerate(md["trcr2prob"]):
        lb, rb = md["trcr2bl2r"][ti], md["trcr2br2l"][ti]
        os_ = l2r_bulk(pt) - 1 if pt & CONSERVE else (0 if pt & NONNEGATIVE else 1)
        r2l[rb, root] = l2r[lb + os_, root]
        if (pt & CONSISTENT) and not (pt & SHAPEPRESERVE):
            r2l[rb + 1, root] = l2r[lb, root]
            r2l[rb + 2, root] = l2r[lb + 2, root]


def _solve(pt, rho_me, rec_me, X, rho0, rec0, rho1, rec1, qlo, qhi, prefer):
    """One node problem on {lo, orig, hi} records, as the kernels call solve_node."""
    t_only = bool(pt & CONSISTENT) and not (pt & SHAPEPRESERVE)
    if t_only:
        pd, k0d, k1d = [qlo, rec_me[1], qhi], [qlo, rec0[1], qhi], [qlo, rec1[1], qhi]
    elif pt & NONNEGATIVE:
        pd, k0d, k1d = [0, 0, 0], [rec0[1], 0, 0], [rec1[1], 0, 0]
    else:
        pd, k0d, k1d = rec_me[:3], rec0[:3], rec1[:3]
    return o.solve_node_problem(pt, rho_me, pd, X, rho0, k0d, rho1, k1d, prefer)


def down_stage(stage, md, l2r, r2l, prefer=False):
    """Mirrors qlt_down_kernel / qlt_down_micro_kernel."""
    for t in range(len(stage["tiles"])):
        in0, nleaf, out, nint, kids = tile_local_nodes(stage, t)
        mic = tile_micro(stage, t)
        nn = nleaf + nint
        rho = np.zeros(nn)
        if mic is None:
            rho[:nleaf] = l2r[0, in0:in0 + nleaf]
        else:
            mrho = []  # per micro: (cell rhos, rho of inner pair t)
            for i, (start, shape) in enumerate(mic):
                c = [l2r[0, in0 + k] for k in micro_cells(start, shape)]
                rt = (c[1] + c[2]) if shape == 3 else (c[0] + c[1]) if shape >= 1 else 0.0
                rho[i] = c[0] if shape == 0 else rt if shape == 1 else (rt + c[2]) if shape == 2 else c[0] + rt
                mrho.append((c, rt))
        for j in range(nint):
            rho[nleaf + j] = rho[kids[j, 0]] + rho[kids[j, 1]]
        outs = tile_outs(stage, t)
        for ti, pt in enumerate(md["trcr2prob"]):
            pt = int(pt)
            lb, rb = md["trcr2bl2r"][ti], md["trcr2br2l"][ti]
            sp = bool(pt & SHAPEPRESERVE)
            t_only = bool(pt & CONSISTENT) and not sp
            of = 1 if (sp or (pt & CONSISTENT)) else 0

            def cell_rec(cell):
                r = np.zeros(3)
                if sp:
                    r[:] = [l2r[lb + f, cell] for f in range(3)]
                else:
                    r[1] = l2r[lb + of, cell]
                return r

            rec = np.zeros((nn, 4))
            if mic is None:
                for i in range(nleaf):
                    rec[i, :3] = cell_rec(in0 + i)
            else:
                mrec = []  # per micro: (cell records, record of inner pair t)
                for i, (start, shape) in enumerate(mic):
                    c = [cell_rec(in0 + k) for k in micro_cells(start, shape)]
                    tr = (c[1] + c[2]) if shape == 3 else (c[0] + c[1]) if shape >= 1 else np.zeros(3)
                    rec[i, :3] = (c[0] if shape == 0 else tr if shape == 1 else (tr + c[2]) if shape == 2
                                  else c[0] + tr)
                    mrec.append((c, tr))
            for j in range(nint):
                rec[nleaf + j, :3] = rec[kids[j, 0], :3] + rec[kids[j, 1], :3]
            for k, lid in enumerate(outs):
                rec[lid, 3] = r2l[rb, out + k]
            qlo = qhi = 0.0
            if t_only:
                qlo, qhi = r2l[rb + 1, out], r2l[rb + 2, out]
            for j in range(nint - 1, -1, -1):
                me, c0, c1 = nleaf + j, kids[j, 0], kids[j, 1]
                rec[c0, 3], rec[c1, 3] = _solve(pt, rho[me], rec[me], rec[me, 3], rho[c0], rec[c0],
                                                rho[c1], rec[c1], qlo, qhi, prefer)
            if mic is None:
                cells = [(in0 + i, rec[i, 3]) for i in range(nleaf)]
            else:
                cells = []
                for i, (start, shape) in enumerate(mic):
                    (c, tr), (cr, rt) = mrec[i], mrho[i]
                    Xr = rec[i, 3]
                    ids = [in0 + k for k in micro_cells(start, shape)]
                    if shape == 0:
                        X = [Xr]
                    elif shape == 1:
                        X = list(_solve(pt, rho[i], rec[i], Xr, cr[0], c[0], cr[1], c[1], qlo, qhi, prefer))
                    elif shape == 2:
                        Xt, Xc = _solve(pt, rho[i], rec[i], Xr, rt, tr, cr[2], c[2], qlo, qhi, prefer)
                        Xa, Xb = _solve(pt, rt, tr, Xt, cr[0], c[0], cr[1], c[1], qlo, qhi, prefer)
                        X = [Xa, Xb, Xc]
                    else:
                        Xa, Xt = _solve(pt, rho[i], rec[i], Xr, cr[0], c[0], rt, tr, qlo, qhi, prefer)
                        Xb, Xc = _solve(pt, rt, tr, Xt, cr[1], c[1], cr[2], c[2], qlo, qhi, prefer)
                        X = [Xa, Xb, Xc]
                    cells += list(zip(ids, X))
            for cell, x in cells:
                r2l[rb, cell] = x
                if t_only:
                    r2l[rb + 1, cell], r2l[rb + 2, cell] = qlo, qhi
                    l2r[lb, cell], l2r[lb + 2, cell] = qlo, qhi


def level_up_nodes(step, data, ops, zero_init=False):
    """Level mode, mirrors level_up_kernel: the nodes a rank owns above its pure subtrees."""
    for me, c0, c1 in step["nodes"]:
        data[:, me] = _comb(ops, data[:, c0], data[:, c1], zero_init)


def level_down_nodes(step, md, l2r, r2l, prefer=False):
    """Mirrors qlt_level_down_kernel."""
    for me, c0, c1 in step["nodes"]:
        rho0, rho1 = l2r[0, c0], l2r[0, c1]
        for ti, pt in enumerate(md["trcr2prob"]):
            pt = int(pt)
            lb, rb = md["trcr2bl2r"][ti], md["trcr2br2l"][ti]
            sp = bool(pt & SHAPEPRESERVE)
            t_only = bool(pt & CONSISTENT) and not sp
            of = 1 if (sp or (pt & CONSISTENT)) else 0
            recs = []
            for c in (c0, c1):
                r = np.zeros(3)
                if sp:
                    r[:] = [l2r[lb + f, c] for f in range(3)]
                else:
                    r[1] = l2r[lb + of, c]
                recs.append(r)
            qlo = qhi = 0.0
            if t_only:
                qlo, qhi = r2l[rb + 1, me], r2l[rb + 2, me]
            x0, x1 = _solve(pt, rho0 + rho1, recs[0] + recs[1], r2l[rb, me], rho0, recs[0], rho1, recs[1],
                            qlo, qhi, prefer)
            for c, x in ((c0, x0), (c1, x1)):
                r2l[rb, c] = x
                if t_only:
                    r2l[rb + 1, c], r2l[rb + 2, c] = qlo, qhi
                    l2r[lb, c], l2r[lb + 2, c] = qlo, qhi


class Mailbox:
    """In-process stand-in for point-to-point messages between rank THREADS: put never blocks,
    get waits. Keys are (src, dst, tag)."""

    def __init__(self):
        import threading
        self.cv = threading.Condition()
        self.box = {}

    def put(self, key, arr):
        with self.cv:
            assert key not in self.box, key
            self.box[key] = np.array(arr, copy=True)
            self.cv.notify_all()

    def get(self, key, timeout=30):
        with self.cv:
            if not self.cv.wait_for(lambda: key in self.box, timeout):
                raise RuntimeError("no message %r" % (key,))
            return self.box.pop(key)


class RankState:
    """One rank's buffers and plan; the exchange is supplied by the caller."""

    def __init__(self, plan, md, prefer=False):
        self.plan, self.md, self.prefer = plan, md, prefer
        self.l2rndps, self.r2lndps = int(md["prob2bl2r"][6]), int(md["prob2br2l"][6])
        S = int(plan["stride"])
        self.l2r = np.zeros((self.l2rndps, S))
        self.r2l = np.zeros((self.r2lndps, S))
        self.ops = field_ops(md, self.l2rndps)
        self.owns_root = True  # level mode: set by the caller (plan["root_owner"] == its rank)

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

    def top_and_down(self, use_apex=False):
        """use_apex: sweep the replicated top as the ONE plain tile the apex kernels use
        (plan["apex"]) instead of the staged top; the results must be the same bits."""
        if use_apex and self.plan.get("apex") is not None:
            ap = self.plan["apex"]
            up_stage(ap, self.l2r, self.ops)
            root_compute(self.md, int(ap["tiles"][0][2]), self.l2r, self.r2l)  # the tile's output slot
            down_stage(ap, self.md, self.l2r, self.r2l, self.prefer)
            tops = []
        else:
            tops = self.plan["top"]
            for st in tops:
                up_stage(st, self.l2r, self.ops)
            root_compute(self.md, int(self.plan["root_slot"]), self.l2r, self.r2l)
        for st in reversed(tops):
            down_stage(st, self.md, self.l2r, self.r2l, self.prefer)
        if len(self.plan["top"]):
            for ts, ls in zip(self.plan["my_top_slots"], self.plan["my_root_slots"]):
                self.r2l[:, ls] = self.r2l[:, ts]
        for st in reversed(self.plan["local"]):
            down_stage(st, self.md, self.l2r, self.r2l, self.prefer)

    # ---- level mode (plan["level_mode"]): xch(tag, sends=[(peer, array)], recvs=[(peer, count)])
    # -> received arrays [count][nfields], in the order of recvs
    def _level_exchange(self, step, down, data, xch):
        send_slots, recv_slots = (step["in_slots"], step["out_slots"]) if down else (step["out_slots"], step["in_slots"])
        send_msgs, recv_msgs = (step["in_msgs"], step["out_msgs"]) if down else (step["out_msgs"], step["in_msgs"])
        if not len(send_msgs) and not len(recv_msgs):
            return
        sends = [(int(p), data[:, send_slots[f:f + c]].T.copy()) for p, f, c in send_msgs]
        got = xch(("down" if down else "up", int(step["level"])), sends, [(int(p), int(c)) for p, f, c in recv_msgs])
        for (p, f, c), a in zip(recv_msgs, got):
            data[:, recv_slots[f:f + c]] = np.asarray(a).reshape(c, data.shape[0]).T

    def run_level_mode(self, xch):
        """QLT::run's level-mode branch after the local leaves->root stages."""
        for step in self.plan["steps"]:
            level_up_nodes(step, self.l2r, self.ops)
            self._level_exchange(step, False, self.l2r, xch)
        if self.owns_root:
            root_compute(self.md, int(self.plan["root_slot"]), self.l2r, self.r2l)
        for step in reversed(self.plan["steps"]):
            self._level_exchange(step, True, self.r2l, xch)
            level_down_nodes(step, self.md, self.l2r, self.r2l, self.prefer)
        for st in reversed(self.plan["local"]):
            down_stage(st, self.md, self.l2r, self.r2l, self.prefer)

    def get(self, n):
        return np.array([self.r2l[self.md["trcr2br2l"][ti], :n] for ti in range(len(self.md["trcr2prob"]))])
