#!/usr/bin/env python3
"""Per-kernel HBM traffic from tools/profile_traffic.sh (rocprofv3 --pmc FETCH_SIZE and
--pmc WRITE_SIZE, separate passes). Prints a table and writes <dir>/r_traffic.json in
the format bench.py reads from profiles/ (copy it there under the round's name)."""
import collections
import csv
import glob
import json
import re
import sys

d = sys.argv[1]
ncells, nt = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1500000, 40)
# rocprof kernel name -> the name bench.py's launch profiler uses, first stage only (largest grid)
OURS = {"qlt_down_micro_kernel": "qlt_down_kernel", "qlt_down_sp_kernel": "qlt_down_kernel",
        "tree_up_micro_kernel": "tree_up_kernel",
        "caas_reduce_kernel": "caas_reduce_kernel", "caas_finish_kernel": "caas_finish_kernel",
        # CAAS fused with the BFB reducer's first stage (bench.py --caas-bfb N): group sums in registers / in LDS
        "caas_reduce_bfb_p2_kernel": "caas_reduce_bfb_kernel", "caas_reduce_bfb_kernel": "caas_reduce_bfb_kernel",
        "caas_reduce_accum_lds_kernel": "caas_reduce_accum_kernel"}
raw = {}
for tag in ("FETCH_SIZE", "WRITE_SIZE", "odd_FETCH_SIZE", "odd_WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (d, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
            if m and m.group(1) in OURS:
                agg[(m.group(1), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        raw.setdefault(k, {})[tag] = sum(v) / len(v)
kernels = {}
print("%-24s %10s %14s %14s %12s" % ("kernel", "grid", "FETCH_SIZE KB", "WRITE_SIZE KB", "HBM bytes"))
for (name, grid), v in sorted(raw.items()):
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue  # only present in the odd-size calibration passes
    hbm = 2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024
    print("%-24s %10d %14.0f %14.0f %12.4g" % (name, grid, v["FETCH_SIZE"], v["WRITE_SIZE"], hbm))
    kernels[OURS[name]] = {"rocprof_name": name, "grid": grid, "FETCH_SIZE_KB": v["FETCH_SIZE"],
                           "WRITE_SIZE_KB": v["WRITE_SIZE"], "hbm_bytes": hbm}
cal = {"%s grid=%d" % k: {t: x for t, x in v.items() if t.startswith("odd_")}
       for k, v in raw.items() if k[0].startswith("caas") and any(t.startswith("odd_") for t in v)}
json.dump({
    "_comment": "HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, "
                "tools/profile_traffic.sh), bench.py --serial on 1 GPU. Counters are in KB. Correction per "
                "MI355X_MICROARCH.md: FETCH_SIZE reports exactly half of streamed read bytes on gfx950 -- checked "
                "here on a known byte count: caas_reduce_kernel reads ncells*ntracers*32 B per launch, and its "
                "FETCH_SIZE is half of that with 16-byte accesses and (odd cell count, see calibration) with 8-byte "
                "accesses alike -- so hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024. WRITE_SIZE is exact.",
    # sha256 of the cedr_kernels.hip the profiled library was built from (KSHA, set by tools/final_profiles.sh from
    # compose_amd.kernels_sha256()): bench.py refuses this file for a library built from other kernel code
    "kernels_sha256": __import__("os").environ.get("KSHA"),
    "config": {"ncells": ncells, "ntracers": nt}, "kernels": kernels, "calibration_odd_cell_count": cal},
    open(d + "/r_traffic.json", "w"), indent=1)
