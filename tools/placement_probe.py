"""Is the first-stage leaves->root kernel's fast / slow state a property of WHERE a QLT's buffers were allocated?
One process: six QLT objects created one after the other and all kept alive (so each gets other device memory), the
kernel timed on each; then again in reverse order."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import compose_amd as ca  # noqa: E402

ncells, nt = 1500000, 40
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
rh = 0.5 * (1 + torch.rand(ncells, dtype=torch.float64, device=dev, generator=g))
qmin = -0.75 + torch.rand((nt, ncells), dtype=torch.float64, device=dev, generator=g)
qmax = qmin + torch.rand((nt, ncells), dtype=torch.float64, device=dev, generator=g)
prev = (qmin + (qmax - qmin) * torch.rand((nt, ncells), dtype=torch.float64, device=dev, generator=g)) * rh
Qm = prev[:, torch.randperm(ncells, device=dev, generator=g)].contiguous()
lo, hi = qmin * rh, qmax * rh
tree = ca.Tree.make_1d(ncells, 1, False, False)


def make():
    q = ca.QLT(ca.Parallel.self_(), ncells, tree)
    for _ in range(nt):
        q.declare_tracer(7, 0)
    q.end_tracer_declarations()
    q.finish_setup()
    q.set_rhom(rh)
    q.set_Qm(Qm, lo, hi, prev)
    return q


def measure(q, steps=25):
    for _ in range(3):
        q.run()
    torch.cuda.synchronize()
    ca.profile_start()
    for _ in range(steps):
        q.run()
    torch.cuda.synchronize()
    acc = {}
    for name, aux, ms in ca.profile_stop():
        a = acc.setdefault((name, aux), [0.0, 0]); a[0] += ms; a[1] += 1
    big = max(aux for (name, aux) in acc if name == "tree_up_kernel")
    return acc[("tree_up_kernel", big)][0] / acc[("tree_up_kernel", big)][1], acc[("qlt_down_kernel", big)][0] / acc[("qlt_down_kernel", big)][1]


qs = []
for i in range(int(os.environ.get("NOBJ", "6"))):
    qs.append(make())
    op = qs[-1].get_device_op()
    print("object %d (others kept): l2r 0x%x (low 32 bits 0x%08x) r2l 0x%x stride %d: up %.4f down %.4f" %
          ((i, op.l2r_data, op.l2r_data & 0xffffffff, op.r2l_data, op.stride) + measure(qs[-1])), flush=True)
for i in reversed(range(len(qs))):
    print("again object %d: up %.4f down %.4f" % ((i,) + measure(qs[i])), flush=True)
