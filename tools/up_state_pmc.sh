#!/bin/bash
# Several fresh processes in a row, each under rocprofv3 with the kernel trace and the address-translation counters:
# does a process whose first-stage leaves->root sweep is slow miss more in the TLBs, or wait longer for the L2?
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/up_state
rm -rf $O; mkdir -p $O
N=${1:-6}
for i in $(seq 1 $N); do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc ${PMC:-TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum} \
    --output-format csv -d $O/p$i -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --serial > $O/p$i.log 2>&1 || { echo "process $i failed"; tail -3 $O/p$i.log; continue; }
  python3 - $O/p$i $i <<'PY'
import csv, glob, sys, collections, re
d, i = sys.argv[1], sys.argv[2]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size", r.get("Grid_Size_X", "")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); t = collections.defaultdict(float)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        m = re.search(r'(tree_up_micro_kernel|qlt_down_sp_kernel|caas_reduce_kernel|caas_finish_kernel)', r["Kernel_Name"])
        if not m: continue
        k = (m.group(1), r["Grid_Size"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); cnt[k] += 1
            if r["Dispatch_Id"] in dur: t[k] += dur[r["Dispatch_Id"]][1]
for k in sorted(agg, key=lambda k: -t[k])[:4]:
    n = cnt[k]; v = agg[k]
    names = sorted(v)
    print("process %s %-22s %9s: %.4f ms | " % (i, k[0], k[1], t[k]/n/1e6) + "  ".join("%s %.4g" % (c.replace("_sum", ""), v[c]/n) for c in names), flush=True)
PY
done
