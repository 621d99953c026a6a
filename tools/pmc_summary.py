#!/usr/bin/env python3
"""Summarise rocprofv3 outputs written by tools/profile_gpu.sh: per-kernel average
duration from the kernel trace stats, and per-dispatch SQ counters."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
for f in glob.glob(d + "/kt/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats:", f)
    for r in list(csv.DictReader(open(f)))[:8]:
        print("  %-60s calls %4s avg_us %10.1f  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
for f in glob.glob(d + "/pmc/**/*counter_collection.csv", recursive=True):
    print("== counters (per dispatch averages):", f)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    ndisp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"], r["VGPR_Count"], r["LDS_Block_Size"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[k].add(r["Dispatch_Id"])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:6]:
        n = len(ndisp[k])
        wc = v["SQ_WAVE_CYCLES"] / n
        print("  %s grid %s vgpr %s lds %s  dispatches %d" % (k[0], k[1], k[2], k[3], n))
        print("     wave_cycles %.3g  active %.0f%%  wait_any %.0f%%  wait_inst %.0f%%  valu %.3g  lds_inst %.3g  lds_conflict/idx_active %.2f" % (
            wc, 100 * v["SQ_ACTIVE_INST_ANY"] / n / wc, 100 * v["SQ_WAIT_ANY"] / n / wc,
            100 * v["SQ_WAIT_INST_ANY"] / n / wc, v["SQ_INSTS_VALU"] / n, v["SQ_INSTS_LDS"] / n,
            v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1)))
