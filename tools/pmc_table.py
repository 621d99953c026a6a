#!/usr/bin/env python3
"""Per (kernel, grid) averages of every counter found under <dir>/pmc_*/ (rocprofv3 --pmc passes).
usage: pmc_table.py <dir> [out.json]"""
import collections, csv, glob, json, re, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/pmc_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        k = (m.group(1) if m else r["Kernel_Name"][:30], int(r["Grid_Size"]), r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""))
        per[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, disp, c), v in per.items():
        agg[k][c].append(v)
out = {}
for k, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    row = {c: sum(v) / len(v) for c, v in cs.items()}
    name = "%s grid=%d vgpr=%s lds=%s" % k
    out[name] = row
    w = row.get("SQ_WAVES", 0) or 1
    print(name)
    print("   " + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(row.items())))
    if "SQ_INSTS_VALU" in row:
        print("   per wave: VALU %.0f SALU %.0f LDS %.0f | lanes/VALU %.1f | VALU active %.1f%% of wave cycles" % (
            row["SQ_INSTS_VALU"] / w, row.get("SQ_INSTS_SALU", 0) / w, row.get("SQ_INSTS_LDS", 0) / w,
            row.get("SQ_THREAD_CYCLES_VALU", 0) / max(row.get("SQ_ACTIVE_INST_VALU", 1), 1),
            100.0 * row.get("SQ_ACTIVE_INST_VALU", 0) / max(row.get("SQ_WAVE_CYCLES", 1), 1)))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
