"""Diagnostic: wall-clock phase stamps of the fused top kernels from a -DCEDR_TOP_STAMPS build
(compose_amd/variants/topstamps.so), at the 187.5k-cell shard: single rank (mode 1) and rank 0 of 8
with a stub all-gather (top-up + mode 2)."""
import ctypes as C, os, sys
os.environ["CEDR_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "compose_amd", "variants", "topstamps.so")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, compose_amd as ca
sys.argv = ["bench"]
import bench
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
L = ca.lib()
L.cedr_hip_debug_top_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
P, n, nt = 8, 1500000, 40
def mk(par, ncells, world):
    q = ca.QLT(par, ncells, ca.Tree.make_1d(ncells, world, False, False))
    for _ in range(nt): q.declare_tracer(7, 0)
    q.end_tracer_declarations(); q.finish_setup(); return q
def ag(s, r, cnt, st):
    for k in range(P): L.cedr_memcpy_d2d_async(r + 8*cnt*k, s, 8*cnt, st)
    return 0
def ar(s, r, cnt, st):
    if s != r: L.cedr_memcpy_d2d_async(r, s, 8*cnt, st)
    return 0
for label, q in (("single rank 187.5k (mode 1)", mk(ca.Parallel.self_(), n//P, 1)),
                 ("rank 0 of 8 (top-up, mode 2)", mk(ca.Parallel.callbacks(0, P, ag, ar), n, P))):
    m = q.nlclcells()
    rhom, Qm, lo, hi, prev = bench.gen_inputs(torch, m, nt, 1, dev)
    q.set_rhom(rhom); q.set_Qm(Qm, lo, hi, prev)
    for _ in range(5): q.run()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    L.cedr_hip_debug_top_stamps(buf)
    t = list(buf)
    print(label)
    names = {0: "start", 1: "level-0 loads + sums", 2: "other sums + reciprocals", 3: "outputs' masses / apex up"}
    prev_t = t[0]
    for i in range(1, 31):
        if t[i] == 0 or t[i] < t[0]: continue
        print("  %-28s +%.2f us" % (names.get(i, "down step %d done" % (i - 4) if i > 4 else "down starts") if i != 30 else "end", (t[i] - prev_t)*0.01))
        prev_t = t[i]
    print("  total %.2f us" % ((t[30] - t[0])*0.01))
    if t[61] > t[60]:
        print("  shader clock over the launch: %.0f MHz" % ((t[61] - t[60])/((t[30] - t[0])*0.01)))
    if t[32]:
        print("  tree_top_up: loads %.2f us, level 0 %.2f us, other levels %.2f us" % (0, (t[33]-t[32])*0.01, (t[34]-t[33])*0.01))
