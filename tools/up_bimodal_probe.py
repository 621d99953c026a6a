"""Probe: does the leaves->root sweep's time (bimodal from process to process) go with where the buffers are?
Prints the device addresses of QLT's buffers and the stage-1 kernel times of one bench-like process."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
import compose_amd as ca
import bench

n, nt = 1500000, 40
dev = torch.device("cuda", 0)
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hold = [torch.empty(pre * (1 << 20), dtype=torch.uint8, device=dev)] if pre else []  # shift later allocations
rhom, Qm, lo, hi, prev = bench.gen_inputs(torch, n, nt, 1234, dev)
q = ca.QLT(ca.Parallel.self_(), n, ca.Tree.make_1d(n, 1))
for _ in range(nt):
    q.declare_tracer(7, 0)
q.end_tracer_declarations()
q.finish_setup()
op = q.get_device_op()
q.set_rhom(rhom); q.set_Qm(Qm, lo, hi, prev)
for _ in range(3):
    q.run()
torch.cuda.synchronize()
ca.profile_start()
for _ in range(20):
    q.run()
recs = ca.profile_stop()
per = {}
for name, aux, ms in recs:
    per.setdefault((name, aux), []).append(ms)
up = sum(per[("tree_up_kernel", 8192)]) / 20
dn = sum(per[("qlt_down_kernel", 8192)]) / 20
print("pre %4d MB  l2r %#x (mod 2MB %#x, mod 1GB %#x)  r2l %#x  stride %d  up %.4f ms  down %.4f ms" %
      (pre, op.l2r_data, op.l2r_data % (2 << 20), op.l2r_data % (1 << 30), op.r2l_data, op.stride, up, dn), flush=True)
