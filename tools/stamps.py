"""Diagnostic: phase shares of the micro down kernel from a -DCEDR_STAMPS build."""
import ctypes as C, os, sys
os.environ["CEDR_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "compose_amd", "variants", os.environ.get("CEDR_STAMPS_LIB", "stamps.so"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, compose_amd as ca
sys.argv = ["bench"]
import bench
n, nt = 1500000, 40
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
q = ca.QLT(ca.Parallel.self_(), n, ca.Tree.make_1d(n))
for _ in range(nt): q.declare_tracer(7, 0)
q.end_tracer_declarations(); q.finish_setup()
rhom, Qm, lo, hi, prev = bench.gen_inputs(torch, n, nt, 1, dev)
q.set_rhom(rhom); q.set_Qm(Qm, lo, hi, prev)
L = ca.lib()
L.cedr_hip_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 64)()
q.run(); torch.cuda.synchronize(); L.cedr_hip_debug_stamps(buf, 1)
for _ in range(3): q.run()
torch.cuda.synchronize(); L.cedr_hip_debug_stamps(buf, 1)
names = ["setup", "loads arrive", "sums", "barrier1", "LDS NPs", "register NPs+stores"]
unames = ["setup+first loads", "issue prefetch+micro combines", "barrier", "LDS levels", "outputs+barrier", "wait prefetch"]
for w in range(4):
    tot = sum(buf[w*8+i] for i in range(6))
    print("down wave", w, "total cycles/run/WG %.0f" % (tot/3/8192), " ".join("%s %.1f%%" % (names[i], 100.0*buf[w*8+i]/tot) for i in range(6)))
for w in range(4):
    tot = sum(buf[32+w*8+i] for i in range(6))
    print("up   wave", w, "total cycles/run/WG %.0f" % (tot/3/8192), " ".join("%s %.1f%%" % (unames[i], 100.0*buf[32+w*8+i]/tot) for i in range(6)))
