import sys, time, ctypes as C
sys.path.insert(0, '/root/repo'); sys.argv=['x']
import torch, compose_amd as ca, bench
dev = torch.device('cuda', 0); torch.zeros(1, device=dev)
L = ca.lib()
P, n, nt = 8, 1500000, 40
def mk(par, ncells, world, prune=None):
    tree = ca.Tree.make_1d(ncells, world, False, False)
    if prune is not None: tree = tree.prune(prune)   # the partial tree a rank passes at scale (Node::level)
    q = ca.QLT(par, ncells, tree)
    for _ in range(nt): q.declare_tracer(7, 0)
    q.end_tracer_declarations(); q.finish_setup(); return q
def ag(s, r, cnt, st):   # every "rank" contributes my data: stream-ordered device copies
    for k in range(P):
        L.cedr_memcpy_d2d_async(r + 8*cnt*k, s, 8*cnt, st)
    return 0
def ar(s, r, cnt, st):
    if s != r: L.cedr_memcpy_d2d_async(r, s, 8*cnt, st)
    return 0
def timeit(f, reps=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/reps*1e3
fake = torch.ones(1 << 16, dtype=torch.float64, device=dev)  # stands in for what peers would send
def xch(ops, st):   # one round of point-to-point messages: receives are stream-ordered device copies
    for peer, ptr, cnt, is_send in ops:
        if not is_send: L.cedr_memcpy_d2d_async(ptr, fake.data_ptr(), 8*cnt, st)
    return 0
qa = mk(ca.Parallel.self_(), n//P, 1)
qb = mk(ca.Parallel.rehearsal(ca.Parallel.self_(), P), n, P)  # rank 0 of 8: collectives = stream-ordered device copies, no Python in the loop
qc = mk(ca.Parallel.callbacks_p2p(0, P, ag, ar, xch), n, P, prune=0)
assert qc.plan()["level_mode"]
for q in (qa, qb, qc):
    m = q.nlclcells()
    rhom, Qm, lo, hi, prev = bench.gen_inputs(torch, m, nt, 1, dev)
    q.set_rhom(rhom); q.set_Qm(Qm, lo, hi, prev)
    st = torch.cuda.Stream(device=dev); q.set_stream(st.cuda_stream)
print('single rank, 187.5k cells : %.4f ms' % timeit(qa.run))
print('rank 0 of 8, 1.5M cells   : %.4f ms' % timeit(qb.run))
print('rank 0 of 8, partial tree (level mode, stub messages): %.4f ms' % timeit(qc.run))
for label, q in (("single rank, 187.5k cells", qa), ("rank 0 of 8, 1.5M cells", qb),
                 ("rank 0 of 8, partial tree (level mode)", qc)):
    ca.profile_start()
    for _ in range(20): q.run()
    recs = ca.profile_stop()
    agg = {}
    for name, aux, ms in recs: agg.setdefault((name, aux), []).append(ms)
    print(label + ", per launch (HIP events):")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])): print('  %-24s %-8d %.4f ms' % (k[0], k[1], sum(v)/len(v)))
