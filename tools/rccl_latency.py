"""Cost of one RCCL collective inside CAAS::run on a size-1 communicator vs the in-process copy."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.argv = ["x"]
import torch, compose_amd as ca, bench
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
n, nt = 187500, 40
def mk(par):
    c = ca.CAAS(par, n)
    for _ in range(nt): c.declare_tracer(7, 0)
    c.end_tracer_declarations(); c.finish_setup()
    rhom, Qm, lo, hi, prev = bench.gen_inputs(torch, n, nt, 1, dev)
    c.set_rhom(rhom); c.set_Qm(Qm, lo, hi, prev); c.set_Qm_source(Qm)
    st = torch.cuda.Stream(device=dev); c.set_stream(st.cuda_stream)
    return c, (rhom, Qm, lo, hi, prev, st)
def timeit(f, reps=300):
    for _ in range(30): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
a, ka = mk(ca.Parallel.self_())
b, kb = mk(ca.Parallel.rccl(ca.Parallel.rccl_unique_id(), 0, 1))
print("CAAS::run 187.5k x 40, self comm : %.4f ms" % timeit(a.run))
print("CAAS::run 187.5k x 40, RCCL size 1: %.4f ms" % timeit(b.run))
