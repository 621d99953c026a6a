"""Probe: CAAS::run over all tracers at once (reduce all, finish all) against the same work in groups of G
tracers (reduce g, finish g, next group), so that a group's arrays may still sit in the 256 MB memory-side cache
when its finish pass re-reads them. Zero-copy mode on slices of the same [tracer][cell] arrays.
Usage: python tools/caas_groups_probe.py [ncells] [ntracers]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
import compose_amd as ca
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
rhom, Qm, lo, hi, prev = bench.gen_inputs(torch, n, nt, 1234, dev)
src = Qm.clone()
par = ca.Parallel.self_()


def make(ntr):
    c = ca.CAAS(par, n)
    for _ in range(ntr):
        c.declare_tracer(7, 0)
    c.end_tracer_declarations()
    c.finish_setup()
    return c


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


for G in (40, 20, 10, 8, 5, 4, 2, 1):
    if nt % G:
        continue
    cs = []
    for g0 in range(0, nt, G):
        c = make(G)
        c.bind_arrays(Qm[g0:g0 + G], lo[g0:g0 + G], hi[g0:g0 + G], prev[g0:g0 + G])
        cs.append(c)

    def run():
        for c in cs:
            c.run()
    Qm.copy_(src)
    ms = timeit(run)
    print("groups of %2d tracers (%3d MB per group): %.4f ms per full pass" % (G, G * n * 32 // 1000000, ms), flush=True)
    del cs
