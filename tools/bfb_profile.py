"""Time BfbTreeAllReducer::allreduce at the production CAAS size (leaves = cells/16, 4*nt fields)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, compose_amd as ca
nleaf, nfield = 93750, 160
b = ca.BfbTreeAllReducer(ca.Parallel.self_(), ca.Tree.make_1d(nleaf), nleaf, nfield)
send = ca.DeviceArray.from_numpy(np.random.default_rng(0).random(nleaf * nfield))
recv = ca.DeviceArray(nfield)
for tr in (False, True):
    for _ in range(5): b.allreduce(send, recv, tr)
    ca.synchronize(); t = time.perf_counter()
    for _ in range(50): b.allreduce(send, recv, tr)
    ca.synchronize(); print("transpose", tr, "%.4f ms per allreduce" % ((time.perf_counter() - t) / 50 * 1e3))
    ca.profile_start()
    for _ in range(10): b.allreduce(send, recv, tr)
    agg = {}
    for name, aux, ms in ca.profile_stop(): agg.setdefault((name, aux), []).append(ms)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])): print("   %-24s %-8d %.4f ms" % (k[0], k[1], sum(v) / len(v)))
