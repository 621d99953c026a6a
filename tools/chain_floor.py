"""Small configurations are bound by a chain of dependent steps, not by HBM: for BASELINE's config B (150k cells x 10 tracers) and for the
187.5k-cell x 40 shard of an 8-GPU run (as rank 0 of 8), every launch of QLT::run with its HIP-event time, the tree levels it walks, and
the FLOOR those levels imply: levels x the measured latency of one dependent level -- 1.12 us per level of node problems root->leaves in
the top tile (a lone wave issues one fp64 instruction per ~7 cycles; profiles/r04_top_stamps.txt), 0.5 us per level of sums leaves->root --
plus ~2.5 us for a launch to start and drain. What is left above the floor is the part a better schedule could still take.
usage: chain_floor.py   (prints; tools/other_configs.sh appends it to profiles/rNN_other_configs.txt)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.argv = ["x"]
import torch, compose_amd as ca, bench
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
L = ca.lib()
DOWN_US, UP_US, LAUNCH_US = 1.12, 0.5, 2.5

def stage_levels(st):
    return max(int(t[6]) for t in st["tiles"]) + (2 if len(st["micro"]) else 0)  # + the two levels inside a micro-subtree

def report(label, par, ncells, world, nt):
    q = ca.QLT(par, ncells, ca.Tree.make_1d(ncells, world, False, False))
    for _ in range(nt): q.declare_tracer(7, 0)
    q.end_tracer_declarations(); q.finish_setup()
    n = q.nlclcells()
    rhom, Qm, lo, hi, prev = bench.gen_inputs(torch, n, nt, 1, dev)
    q.set_rhom(rhom); q.set_Qm(Qm, lo, hi, prev)
    for _ in range(20): q.run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200): q.run()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t)/200*1e3
    ca.profile_start()
    for _ in range(20): q.run()
    agg, order = {}, []
    for name, aux, m in ca.profile_stop():
        if (name, aux) not in agg: order.append((name, aux))
        agg.setdefault((name, aux), []).append(m)
    plan = q.plan()
    lv = [stage_levels(st) for st in plan["local"]]
    apex = 3 if world > 1 else 0  # log2(8) levels above the rank boundaries, swept inside the top launch
    nlaunch = sum(len(v)/20.0 for v in agg.values())
    up_levels = sum(lv[:-1]) if plan["local"][-1]["top"] else sum(lv)
    down_levels = sum(lv) + apex
    top_levels = lv[-1] if plan["local"][-1]["top"] else 0
    floor = (up_levels*UP_US + (top_levels + apex)*UP_US*(1 if world > 1 else 0) + down_levels*DOWN_US + nlaunch*LAUNCH_US)*1e-3
    print("%s: QLT::run %.4f ms; %d launches; tree levels per stage %s%s" %
          (label, ms, round(nlaunch), lv, (" + %d apex levels" % apex) if apex else ""))
    for k in order:
        print("    %-22s %-8d %.4f ms" % (k[0], k[1], sum(agg[k])/len(agg[k])))
    print("    chain floor: %d levels of sums x %.2f us + %d levels of node problems x %.2f us + %d launches x %.1f us = %.4f ms (%.0f %% of the measured time)"
          % (up_levels + ((top_levels + apex) if world > 1 else 0), UP_US, down_levels, DOWN_US, round(nlaunch), LAUNCH_US, floor, 100*floor/ms))
    alg = n*nt*(64.0 + 16.0/nt)
    print("    HBM side: %.1f MB algorithmic = %.4f ms at 8 TB/s (%.1f %% of the measured time)" % (alg/1e6, alg/8e12*1e3, 100*alg/8e12*1e3/ms))

P = 8
def ag(s, r, cnt, st):
    for k in range(P): L.cedr_memcpy_d2d_async(r + 8*cnt*k, s, 8*cnt, st)
    return 0
def ar(s, r, cnt, st):
    if s != r: L.cedr_memcpy_d2d_async(r, s, 8*cnt, st)
    return 0
report("config B (150k cells x 10 tracers, one rank)", ca.Parallel.self_(), 150000, 1, 10)
report("187.5k-cell shard x 40 tracers as rank 0 of 8 (rehearsal communicator)", ca.Parallel.rehearsal(ca.Parallel.self_(), P), 1500000, P, 40)
