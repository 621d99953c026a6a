#!/usr/bin/env python3
"""Per (kernel, grid) durations from a rocprofv3 --kernel-trace CSV.

rocprofv3 --stats averages by kernel NAME; since every stage of the tree sweeps runs
the same micro kernels (8192 tiles down to 1), that average mixes the stage bench.py's
roofline is about with the tiny ones. This splits the same trace by launch grid so the
stage-1 rows can be compared with bench.py's HIP-event averages.
usage: kernel_stats_by_grid.py <kernel_trace.csv> <out.csv>"""
import collections
import csv
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(src)):
    m = re.search(r"(\w+_kernel)(<[^>(]*>)?", r["Kernel_Name"])
    if not m or not m.group(1).startswith(("qlt_", "tree_", "caas_", "gather_slots", "scatter_slots",
                                            "copy_slots")):
        continue
    grid = "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    wg = int(r["Workgroup_Size_X"])
    # template arguments kept: the zero-copy (ROWS = true) builds of the sweeps run with the same grid as the packed
    # ones in bench.py's pack/unpack leg and must not be averaged into the rows the roofline is about
    agg[(m.group(1) + (m.group(2) or ""), grid, wg, r["VGPR_Count"], r["LDS_Block_Size"])].append(
        int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Grid(threads)", "Workgroups_X", "VGPRs", "LDS_bytes", "Calls", "TotalNs", "AverageNs",
                "MinNs", "MaxNs"])
    for (name, grid, wg, vgpr, lds), v in rows:
        gx = int(grid.split("x")[0]) // wg
        w.writerow([name, grid, gx, vgpr, lds, len(v), sum(v), "%.1f" % (sum(v) / len(v)), min(v), max(v)])
try:
    for (name, grid, wg, vgpr, lds), v in rows:
        gx = int(grid.split("x")[0]) // wg
        print("%-28s grid %-18s wgs_x %-6d calls %4d avg %10.1f us" % (name, grid, gx, len(v), sum(v) / len(v) / 1e3))
except BrokenPipeError:  # `| head`
    pass
