#!/bin/bash
# Run on the GPU box (through gpurun): dynamic instruction counts of the QLT kernels.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS \
  --output-format csv -d "$R/gpurun_out/prof_valu" -- \
  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial > gpurun_out/prof_valu.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/prof_valu/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        import re as _re
        _m = _re.search(r"(\w+_kernel)", r["Kernel_Name"])
        k = (_m.group(1) if _m else r["Kernel_Name"][:30], r["Grid_Size"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:6]:
        n = len(nd[k]); w = v["SQ_WAVES"]/n
        print(k, "disp", n, "waves %.3g" % w)
        print("   per wave: valu %.0f salu %.0f lds %.0f | active_valu_cycles/wave %.0f thread_cycles_valu/active_valu (lanes) %.1f | wave_cycles/wave %.0f busy_cycles %.3g" % (
            v["SQ_INSTS_VALU"]/n/w, v["SQ_INSTS_SALU"]/n/w, v["SQ_INSTS_LDS"]/n/w, v["SQ_ACTIVE_INST_VALU"]/n/w,
            v["SQ_THREAD_CYCLES_VALU"]/max(v["SQ_ACTIVE_INST_VALU"],1), v["SQ_WAVE_CYCLES"]/n/w, v["SQ_BUSY_CYCLES"]/n))
PY
