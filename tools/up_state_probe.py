"""What moves the first-stage sweeps' times by 5-12 % between processes and boxes? One process, 1.5M cells x 40 cst
tracers: (1) the same CDR objects timed in segments over ~20 s of continuous running (a state of the machine:
clocks, power, temperature?), (2) the CDRs destroyed and re-created (fresh device buffers: where they land?),
(3) the same after the device heap was fragmented (64 x 1 GB allocated, every other one freed)."""
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


def make():
    tree = ca.Tree.make_1d(ncells, 1, False, False)
    q = ca.QLT(ca.Parallel.self_(), ncells, tree)
    c = ca.CAAS(ca.Parallel.self_(), ncells)
    for x in (q, c):
        for _ in range(nt):
            x.declare_tracer(CST, 0)
        x.end_tracer_declarations()
        x.finish_setup()
    g = torch.Generator(device=dev); g.manual_seed(1)
    rh = 0.5 * (1 + torch.rand(ncells, dtype=torch.float64, device=dev, generator=g))
    qmin = -0.75 + torch.rand((nt, ncells), dtype=torch.float64, device=dev, generator=g)
    qmax = qmin + torch.rand((nt, ncells), dtype=torch.float64, device=dev, generator=g)
    qq = qmin + (qmax - qmin) * torch.rand((nt, ncells), dtype=torch.float64, device=dev, generator=g)
    prev = qq * rh
    perm = torch.randperm(ncells, device=dev, generator=g)
    Qm = prev[:, perm].contiguous()
    for x in (q, c):
        x.set_rhom(rh)
        x.set_Qm(Qm, qmin * rh, qmax * rh, prev)
    c.set_Qm_source(Qm)
    torch.cuda.synchronize()
    return q, c, (rh, qmin, qmax, prev, Qm)


def segment(q, c, steps=40):
    ca.profile_start()
    t0 = time.perf_counter()
    for _ in range(steps):
        q.run(); c.run()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    acc = {}
    for name, aux, ms in ca.profile_stop():
        a = acc.setdefault((name, aux), [0.0, 0])
        a[0] += ms; a[1] += 1
    big = max(aux for (name, aux) in acc if name == "tree_up_kernel")
    f = lambda n, a: acc[(n, a)][0] / acc[(n, a)][1]  # noqa: E731
    caas = sum(v[0] for (n, a), v in acc.items() if n.startswith("caas")) / steps
    return "up %.4f down %.4f caas kernels %.4f step (wall) %.4f" % (f("tree_up_kernel", big), f("qlt_down_kernel", big), caas, wall)


def main():
    t00 = time.time()
    q, c, keep = make()
    for i in range(3):
        q.run(); c.run()
    torch.cuda.synchronize()
    for seg in range(int(os.environ.get("SEGS", "14"))):
        print("t=%5.1fs same objects, segment %2d: %s" % (time.time() - t00, seg, segment(q, c, int(os.environ.get("STEPS", "600")))), flush=True)
    print("idle 5 s"); time.sleep(5)
    for seg in range(3):
        print("t=%5.1fs after the pause, segment %2d: %s" % (time.time() - t00, seg, segment(q, c, 100)), flush=True)
    for rep in range(3):
        del q, c, keep
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        q, c, keep = make()
        q.run(); c.run()
        print("t=%5.1fs re-created %d: %s" % (time.time() - t00, rep, segment(q, c, 100)), flush=True)
    del q, c, keep
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    junk = [torch.empty(1 << 30, dtype=torch.uint8, device=dev) for _ in range(64)]
    junk = junk[::2]
    torch.cuda.synchronize()
    for rep in range(2):
        q, c, keep = make()
        q.run(); c.run()
        print("t=%5.1fs fragmented heap, re-created %d: %s" % (time.time() - t00, rep, segment(q, c, 100)), flush=True)
        del q, c, keep
        torch.cuda.synchronize()


main()
