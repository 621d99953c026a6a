"""Follow-up to placement_probe.py: the QLT's buffers at CHOSEN addresses inside one big allocation (set_buffers):
does the first-stage kernels' time follow the address (offset within the allocation), and at what granularity?"""
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
GB = 1 << 30
arena = torch.empty(48 * GB // 8, dtype=torch.float64, device=dev)
print("arena at 0x%x" % arena.data_ptr())


def run_at(off_bytes):
    q = ca.QLT(ca.Parallel.self_(), ncells, tree)
    for _ in range(nt):
        q.declare_tracer(7, 0)
    q.end_tracer_declarations()
    b1, b2 = q.get_buffers_sizes_tight()
    o = off_bytes // 8
    buf1 = arena[o:o + b1]
    o2 = o + ((b1 + 31) // 32) * 32
    buf2 = arena[o2:o2 + b2]
    q.set_buffers(buf1, buf2)
    q.finish_setup()
    q.set_rhom(rh)
    q.set_Qm(Qm, lo, hi, prev)
    for _ in range(3):
        q.run()
    torch.cuda.synchronize()
    ca.profile_start()
    for _ in range(20):
        q.run()
    torch.cuda.synchronize()
    acc = {}
    for name, aux, ms in ca.profile_stop():
        a = acc.setdefault((name, aux), [0.0, 0]); a[0] += ms; a[1] += 1
    big = max(aux for (name, aux) in acc if name == "tree_up_kernel")
    up = acc[("tree_up_kernel", big)][0] / acc[("tree_up_kernel", big)][1]
    dn = acc[("qlt_down_kernel", big)][0] / acc[("qlt_down_kernel", big)][1]
    return up, dn, (b1 + b2) * 8


offs = [0, 4 * GB, 8 * GB, 12 * GB, 16 * GB, 20 * GB, 24 * GB, 28 * GB, 32 * GB, 0,
        64 << 10, 1 << 20, 2 << 20, 16 << 20, 256 << 20, 1 * GB, 2 * GB, 3 * GB, 5 * GB, 6 * GB, 4 * GB]
for off in offs:
    up, dn, sz = run_at(off)
    print("offset %7.3f GB (%5.2f GB used): up %.4f down %.4f sum %.4f" % (off / GB, sz / GB, up, dn, up + dn), flush=True)
