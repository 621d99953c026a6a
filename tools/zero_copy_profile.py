"""Per-kernel times of QLT::run in packed and in zero-copy (bind_arrays) mode."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.argv = ["x"]
import torch, compose_amd as ca, bench
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
n, nt = 1500000, 40
q = ca.QLT(ca.Parallel.self_(), n, ca.Tree.make_1d(n))
for _ in range(nt): q.declare_tracer(7, 0)
q.end_tracer_declarations(); q.finish_setup()
rhom, Qm, lo, hi, prev = bench.gen_inputs(torch, n, nt, 1, dev)
out = torch.empty_like(Qm)
q.set_rhom(rhom); q.set_Qm(Qm, lo, hi, prev)
def prof(label):
    for _ in range(3): q.run()
    ca.synchronize(); ca.profile_start()
    for _ in range(10): q.run()
    agg = {}
    for name, aux, ms in ca.profile_stop(): agg.setdefault((name, aux), []).append(ms)
    print(label, " ".join("%s[%d]=%.4f" % (k[0].replace("_kernel", ""), k[1], sum(v)/len(v)) for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:4]))
prof("packed   ")
q.bind_arrays(rhom, Qm, lo, hi, prev, out); prof("zero-copy")
q.bind_arrays(rhom, Qm, lo, hi, prev, Qm.clone()); prof("zc, out=copy of Qm")
