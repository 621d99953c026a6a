#!/bin/bash
# Run on the GPU box (through gpurun): TLB / memory-latency counters of the streaming kernels.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
# (more than four TCP counters in one pass exceed the hardware: rocprofv3 aborts and hangs)
timeout -k 10 200 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum \
  --output-format csv -d "$R/gpurun_out/prof_tlb" -- \
  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial > gpurun_out/prof_tlb.log 2>&1
python3 - <<'PY'
import csv, glob, collections, re
for f in glob.glob("gpurun_out/prof_tlb/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        m = re.search(r'(\w+_kernel)', r["Kernel_Name"])
        k = (m.group(1) if m else r["Kernel_Name"][:30], r["Grid_Size"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("TCP_TCC_READ_REQ_sum", 0))[:8]:
        n = len(nd[k])
        print(k, "disp", n)
        print("   utcl1 req %.3g miss %.3g (%.2f%%) hit %.3g | tcc read req %.3g avg latency %.0f cycles | pending stall %.3g gate_en1 %.3g" % (
            v["TCP_UTCL1_REQUEST_sum"]/n, v["TCP_UTCL1_TRANSLATION_MISS_sum"]/n,
            100*v["TCP_UTCL1_TRANSLATION_MISS_sum"]/max(v["TCP_UTCL1_REQUEST_sum"],1), v["TCP_UTCL1_TRANSLATION_HIT_sum"]/n,
            v["TCP_TCC_READ_REQ_sum"]/n, v["TCP_TCC_READ_REQ_LATENCY_sum"]/max(v["TCP_TCC_READ_REQ_sum"],1),
            v["TCP_PENDING_STALL_CYCLES_sum"]/n, v["TCP_GATE_EN1_sum"]/n))
PY
