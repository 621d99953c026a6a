#!/usr/bin/env python3
"""Round 5: does replaying a captured hipGraph of one step (QLT::run then CAAS::run on one stream) beat enqueueing the
step's launches one by one? Two regimes: back to back (the host runs ahead of the GPU: what bench.py times) and with a
host synchronisation after every step (an application that reads results every step). For the small configurations,
where a step is 8 launches of 10-60 us each. Checks that replay gives the same bits as the plain step.
usage: graph_replay.py [NCELLS NTRACERS]..."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import compose_amd as ca  # noqa: E402


def case(n, nt, steps=300):
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

    def step():
        qlt.run()
        caas.run()

    def results():
        a, b = torch.empty_like(Qm), torch.empty_like(Qm)
        qlt.get_Qm(a)
        caas.get_Qm(b)
        torch.cuda.synchronize()
        return a.clone(), b.clone()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    want = results()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            step()
    except Exception as e:  # noqa: BLE001
        print("%d x %d: capture failed: %s" % (n, nt, str(e)[:300]))
        return
    g.replay()
    torch.cuda.synchronize()
    got = results()
    same = bool(torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]))

    def timed(fn, sync_each):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
            if sync_each:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    rows = []
    for rep in range(2):
        rows.append((timed(step, False), timed(g.replay, False), timed(step, True), timed(g.replay, True)))
    for r in rows:
        print("%8d cells x %3d tracers: back to back  launches %.4f ms  graph %.4f ms | sync every step  launches %.4f ms  graph %.4f ms | same bits %s"
              % (n, nt, r[0], r[1], r[2], r[3], same), flush=True)


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]] or [150000, 10, 187500, 40, 1500000, 40]
    print("# kernels_sha256 %s" % ca.kernels_sha256())
    for i in range(0, len(a), 2):
        case(a[i], a[i + 1])
