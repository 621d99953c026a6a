"""One line per configuration: CAAS::run with the BFB reducer fused (the reducer object as UserAllReducer) at 1.5M cells x 40
tracers on one GPU and as rank 0 of 8 (stub transport), for A/B runs of launch-shape knobs (environment variables are read once
per process, so every setting is its own process). usage: caas_bfb_quick.py [accum]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
acc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sys.argv = ["x"]
import torch, compose_amd as ca, bench
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
L = ca.lib()
P, N, nt = 8, 1500000, 40
def ag(s, r, cnt, st):
    for k in range(P): L.cedr_memcpy_d2d_async(r + 8*cnt*k, s, 8*cnt, st)
    return 0
def ar(s, r, cnt, st):
    if s != r: L.cedr_memcpy_d2d_async(r, s, 8*cnt, st)
    return 0
own = os.environ.get("OWN") == "1"  # CAAS with its own all-reduce instead (the plain reduce kernel)
def run(par, world):
    n = N // world
    if own:
        caas = ca.CAAS(par, n)
    else:
        bfb = ca.BfbTreeAllReducer(par, ca.Tree.make_1d(N // acc, world, False, False), N // acc, 4*nt)
        caas = ca.CAAS(par, n, bfb, acc)
    for _ in range(nt): caas.declare_tracer(7, 0)
    caas.end_tracer_declarations(); caas.finish_setup()
    rhom, Qm, lo, hi, prev = bench.gen_inputs(torch, n, nt, 1, dev)
    caas.set_rhom(rhom); caas.set_Qm(Qm, lo, hi, prev); caas.set_Qm_source(Qm)
    for _ in range(10): caas.run()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): caas.run()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t)/50*1e3)
    ca.profile_start()
    for _ in range(20): caas.run()
    ks = {}
    for name, aux, m in ca.profile_stop(): ks.setdefault(name, []).append(m)
    return best, caas.fused_with_bfb(), {k: sum(v)/20 for k, v in ks.items()}
knobs = " ".join("%s=%s" % kv for kv in sorted(os.environ.items()) if kv[0].startswith("CEDR_HIP_"))
for label, par, world in (("1.5M", ca.Parallel.self_(), 1), ("rank0of8", ca.Parallel.rehearsal(ca.Parallel.self_(), P), P)):
    ms, fused, ks = run(par, world)
    print("%-28s acc %d %-9s fused=%d CAAS::run %.4f ms | %s" % (knobs, acc, label, fused, ms,
          " ".join("%s %.4f" % (k.replace("_kernel", ""), v) for k, v in sorted(ks.items(), key=lambda kv: -kv[1]))), flush=True)
