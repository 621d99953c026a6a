"""The production pairing: CAAS::run with the BfbTreeAllReducer as its UserAllReducer
(reference cedr_caas.cpp:137-168,263-266, cedr_bfb_tree_allreduce.cpp:79-159) at 1.5M cells x 40 tracers on
one GPU and at the 187.5k-cell shard (rank 0 of 8, stub transport). Times CAAS::run, lists every launch
(HIP events), and prices the run against the HBM roofline with the algorithmic bytes
    reduce (clip + group sums): 40 B per cell-tracer + 32/accum B (the 4 partial sums per group it writes)
    the reducer's leaves->root sweep: reads those partials once: 32/accum B per cell-tracer
    finish: 24 B per cell-tracer
Round 5: the reducer object itself is CAAS's UserAllReducer (caas::BfbUserAllReducer) -- run() fuses the reduce
pass with the first stage of the reducer's tree, no partial-sum buffer: 64 B per cell-tracer algorithmic -- next to the
same pairing through a callback (the two separate launches: 64 + 64/accum B).
Writes gpurun_out/<RN>_caas_bfb.json and .csv (RN from the environment, default r05; copy to profiles/)."""
import json, os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.argv = ["x"]
import torch, compose_amd as ca, bench
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
L = ca.lib()
P, N, nt = 8, 1500000, 40
PEAK = 8000.0

def ag(s, r, cnt, st):
    for k in range(P): L.cedr_memcpy_d2d_async(r + 8*cnt*k, s, 8*cnt, st)
    return 0
def ar(s, r, cnt, st):
    if s != r: L.cedr_memcpy_d2d_async(r, s, 8*cnt, st)
    return 0

def run(label, par_fn, ncells_global, world, acc, typed=True):
    par = par_fn()
    n = ncells_global // world
    btree = ca.Tree.make_1d(ncells_global // acc, world, False, False)
    bfb = ca.BfbTreeAllReducer(par, btree, ncells_global // acc, 4*nt)
    caas = ca.CAAS(par, n, bfb if typed else (lambda s, r, nl, nf: bfb.allreduce(s, r, True)), acc)
    for _ in range(nt): caas.declare_tracer(7, 0)
    caas.end_tracer_declarations(); caas.finish_setup()
    fused = caas.fused_with_bfb()
    label += ", fused" if fused else ", two launches"
    rhom, Qm, lo, hi, prev = bench.gen_inputs(torch, n, nt, 1, dev)
    caas.set_rhom(rhom); caas.set_Qm(Qm, lo, hi, prev); caas.set_Qm_source(Qm)
    for _ in range(10): caas.run()
    torch.cuda.synchronize(); t = time.perf_counter()
    reps = 100
    for _ in range(reps): caas.run()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t)/reps*1e3
    ca.profile_start()
    for _ in range(20): caas.run()
    agg = {}
    for name, aux, m in ca.profile_stop(): agg.setdefault((name, aux), []).append(m)
    rows = [dict(kernel=k[0], launch=k[1], launches_per_run=len(v)/20.0, avg_ms=sum(v)/len(v)) for k, v in
            sorted(agg.items(), key=lambda kv: -sum(kv[1]))]
    alg = n*nt*(64.0 + (0.0 if fused else 64.0/acc))
    out = dict(config=label, fused=fused, cells_per_gpu=n, ntracers=nt, cells_per_leaf=acc, caas_run_ms=ms,
               updates_per_s=n*nt/(ms*1e-3), algorithmic_bytes=alg, achieved_GBs=alg/(ms*1e-3)/1e9,
               frac_of_8TBs=alg/(ms*1e-3)/1e9/PEAK, launches=rows,
               sum_of_launches_ms=sum(r["avg_ms"]*r["launches_per_run"] for r in rows))
    print("%-60s CAAS::run %.4f ms  %.3e updates/s  %.0f GB/s = %.1f%% of 8 TB/s" %
          (label, ms, out["updates_per_s"], out["achieved_GBs"], 100*out["frac_of_8TBs"]))
    for r in rows: print("    %-28s %-8d x%.0f  %.4f ms" % (r["kernel"], r["launch"], r["launches_per_run"], r["avg_ms"]))
    return out

res = []
accs1 = [int(a) for a in os.environ.get("ACCS1", "4,16").split(",")]
accs8 = [int(a) for a in os.environ.get("ACCS8", "4,20").split(",")]  # (187 500 cells per rank are not a multiple of 16)
for typed in (True, False):
    for acc in accs1:
        res.append(run("1 GPU, 1.5M cells, %d cells per leaf" % acc, ca.Parallel.self_, N, 1, acc, typed))
    for acc in accs8:
        res.append(run("rank 0 of 8 (rehearsal communicator), 187.5k cells, %d cells per leaf" % acc,
                       lambda: ca.Parallel.rehearsal(ca.Parallel.self_(), P), N, P, acc, typed))
# the own-reducer CAAS beside it
def own(label, n):
    caas = ca.CAAS(ca.Parallel.self_(), n)
    for _ in range(nt): caas.declare_tracer(7, 0)
    caas.end_tracer_declarations(); caas.finish_setup()
    rhom, Qm, lo, hi, prev = bench.gen_inputs(torch, n, nt, 1, dev)
    caas.set_rhom(rhom); caas.set_Qm(Qm, lo, hi, prev); caas.set_Qm_source(Qm)
    for _ in range(10): caas.run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(100): caas.run()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t)/100*1e3
    alg = n*nt*64.0
    print("%-60s CAAS::run %.4f ms  %.0f GB/s = %.1f%% of 8 TB/s" % (label, ms, alg/(ms*1e-3)/1e9, 100*alg/(ms*1e-3)/1e9/PEAK))
    return dict(config=label, cells_per_gpu=n, ntracers=nt, caas_run_ms=ms, algorithmic_bytes=alg,
                frac_of_8TBs=alg/(ms*1e-3)/1e9/PEAK)
res.append(own("1 GPU, 1.5M cells, own all-reduce", N))
res.append(own("1 GPU, 187.5k cells, own all-reduce", N//P))
try:
    head = " ".join(open(os.path.join(ROOT, ".head_sha")).read().split())  # written by tools/gpu.sh
except OSError:
    head = "unknown"
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(dict(head=head, kernels_sha256=ca.kernels_sha256(), peak_GBs=PEAK, runs=res), open(os.path.join(ROOT, "gpurun_out", os.environ.get("RN", "r05") + "_caas_bfb.json"), "w"), indent=1)
with open(os.path.join(ROOT, "gpurun_out", os.environ.get("RN", "r05") + "_caas_bfb.csv"), "w") as f:
    f.write("# HEAD %s kernels_sha256 %s\n" % (head, ca.kernels_sha256()))
    f.write("config,kernel,launch,launches_per_run,avg_ms\n")
    for r in res:
        for k in r.get("launches", []):
            f.write("%s,%s,%d,%.1f,%.5f\n" % (r["config"].replace(",", ";"), k["kernel"], k["launch"], k["launches_per_run"], k["avg_ms"]))
