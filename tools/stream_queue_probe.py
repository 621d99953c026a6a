"""Does the first-stage leaves->root kernel's fast / slow state (profiles/r03_experiments.md 11) follow the HIP stream,
i.e. the hardware queue a stream is mapped to? One process, 1.5M cells x 40 cst tracers: QLT::run on the null stream and
on eight streams created one after the other (ROCclr deals streams over its hardware queues in turn), three passes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import compose_amd as ca  # noqa: E402

CST = 7
ncells, nt = 1500000, 40
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
tree = ca.Tree.make_1d(ncells, 1, False, False)
q = ca.QLT(ca.Parallel.self_(), ncells, tree)
for _ in range(nt):
    q.declare_tracer(CST, 0)
q.end_tracer_declarations()
q.finish_setup()
g = torch.Generator(device=dev); g.manual_seed(1)
rh = 0.5 * (1 + torch.rand(ncells, dtype=torch.float64, device=dev, generator=g))
qmin = -0.75 + torch.rand((nt, ncells), dtype=torch.float64, device=dev, generator=g)
qmax = qmin + torch.rand((nt, ncells), dtype=torch.float64, device=dev, generator=g)
prev = (qmin + (qmax - qmin) * torch.rand((nt, ncells), dtype=torch.float64, device=dev, generator=g)) * rh
Qm = prev[:, torch.randperm(ncells, device=dev, generator=g)].contiguous()
q.set_rhom(rh)
q.set_Qm(Qm, qmin * rh, qmax * rh, prev)
torch.cuda.synchronize()


def measure(steps=30):
    ca.profile_start()
    for _ in range(steps):
        q.run()
    torch.cuda.synchronize()
    acc = {}
    for name, aux, ms in ca.profile_stop():
        a = acc.setdefault((name, aux), [0.0, 0]); a[0] += ms; a[1] += 1
    big = max(aux for (name, aux) in acc if name == "tree_up_kernel")
    return acc[("tree_up_kernel", big)][0] / acc[("tree_up_kernel", big)][1], acc[("qlt_down_kernel", big)][0] / acc[("qlt_down_kernel", big)][1]


streams = [None] + [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("NSTREAMS", "8")))]
for p in range(3):
    for i, st in enumerate(streams):
        q.set_stream(st.cuda_stream if st is not None else 0)
        for _ in range(3):
            q.run()
        torch.cuda.synchronize()
        up, down = measure()
        print("pass %d stream %d (%s): up %.4f down %.4f" % (p, i, "null" if st is None else hex(st.cuda_stream), up, down), flush=True)
