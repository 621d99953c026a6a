#!/usr/bin/env python3
"""Which part of a replayed step differs from the plain step at 1.5M cells x 40 tracers (graph_replay.py: 'same bits False')?
QLT and CAAS separately; plain vs plain, replay vs replay, replay vs plain; sizes in between."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import compose_amd as ca  # noqa: E402


def case(n, nt):
    dev = torch.device("cuda", 0)
    par = ca.Parallel.self_()
    qlt = ca.QLT(par, n, ca.Tree.make_1d(n, 1, False, False))
    caas = ca.CAAS(par, n)
    for x in (qlt, caas):
        for _ in range(nt):
            x.declare_tracer(7, 0)
        x.end_tracer_declarations()
        x.finish_setup()
    rhom, Qm, lo, hi, prev = bench.gen_inputs(torch, n, nt, 4242, dev)
    for x in (qlt, caas):
        x.set_rhom(rhom)
        x.set_Qm(Qm, lo, hi, prev)
    caas.set_Qm_source(Qm)
    torch.cuda.synchronize()  # the set_* calls above ran on the null stream; s below does not wait for it
    s = torch.cuda.Stream(device=dev)
    qlt.set_stream(s.cuda_stream)
    caas.set_stream(s.cuda_stream)

    def res(x):
        a = torch.empty_like(Qm)
        x.get_Qm(a)
        torch.cuda.synchronize()
        return a.clone()

    for name, x in (("QLT", qlt), ("CAAS", caas)):
        x.run(); torch.cuda.synchronize(); p1 = res(x)
        x.run(); torch.cuda.synchronize(); p2 = res(x)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            x.run()
        g.replay(); torch.cuda.synchronize(); r1 = res(x)
        g.replay(); torch.cuda.synchronize(); r2 = res(x)
        x.run(); torch.cuda.synchronize(); p3 = res(x)
        d = (r1 - p1).abs().max().item()
        nd = int((r1 != p1).sum().item())
        rows = (r1 != p1).any(dim=1).nonzero().flatten().tolist()[:8]
        print("%8d x %3d %-4s plain==plain %s  replay==replay %s  replay==plain %s (max |diff| %.3e, %d values, tracers %s)  plain after replay==plain %s"
              % (n, nt, name, torch.equal(p1, p2), torch.equal(r1, r2), torch.equal(r1, p1), d, nd, rows, torch.equal(p3, p1)), flush=True)
        if nd:
            cols = (r1 != p1).any(dim=0).nonzero().flatten()
            print("      differing cells: %d, first %s last %s" % (cols.numel(), cols[:6].tolist(), cols[-6:].tolist()), flush=True)


if __name__ == "__main__":
    print("# kernels_sha256 %s" % ca.kernels_sha256())
    for n, nt in ((187500, 40), (750000, 40), (1500000, 10), (1500000, 40)):
        case(n, nt)
