#!/bin/bash
# Round 3: QLT's own buffers as physically contiguous device memory (hipDeviceMallocContiguous) against ordinary hipMalloc.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out/r03env
for pass in 1 2 3 4; do
for cfg in "0 1" "1 1" "0 4" "1 4"; do
  set -- $cfg
  CEDR_HIP_PLACE_VERBOSE=1 CEDR_HIP_CONTIG=$1 CEDR_HIP_PLACE_TRIES=$2 timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --serial --check > gpurun_out/r03env/ct.json 2> gpurun_out/r03env/ct.err || { tail -3 gpurun_out/r03env/ct.err; continue; }
  python3 - "$cfg" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r03env/ct.json"))
k={ (x["kernel"],x["launch"]):x["avg_ms"] for x in d["kernels"] }
nt=max(b for (a,b) in k if a=="qlt_down_kernel")
print("contig,tries %s: qlt %.4f  up %.4f down %.4f  sum %.4f step %.4f check %s" % (sys.argv[1], d["qlt"]["ms_per_run"], k[("tree_up_kernel",nt)], k[("qlt_down_kernel",nt)], k[("tree_up_kernel",nt)]+k[("qlt_down_kernel",nt)], d["ms_per_step"], d.get("check_ok")))
PY
done
done
