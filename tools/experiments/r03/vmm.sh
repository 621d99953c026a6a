#!/bin/bash
# Round 3: QLT's own l2r / r2l buffers through the VMM API (one physical allocation, 1 GB-aligned range) against hipMalloc.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out/r03env
for pass in 1 2 3 4; do
for cfg in "0 1" "0 4"; do
  set -- $cfg
  CEDR_HIP_PLACE_VERBOSE=1 CEDR_HIP_VMM=$1 CEDR_HIP_PLACE_TRIES=$2 timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --serial --check > gpurun_out/r03env/vm.json 2> gpurun_out/r03env/vm.err || { tail -3 gpurun_out/r03env/vm.err; continue; }
  python3 - "$cfg" "$(grep -c 'vmm 1' gpurun_out/r03env/vm.err)" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r03env/vm.json"))
k={ (x["kernel"],x["launch"]):x["avg_ms"] for x in d["kernels"] }
nt=max(b for (a,b) in k if a=="qlt_down_kernel")
print("vmm,tries %s (vmm in use: %s): qlt %.4f  up %.4f down %.4f  sum %.4f step %.4f check %s" % (sys.argv[1], sys.argv[2], d["qlt"]["ms_per_run"], k[("tree_up_kernel",nt)], k[("qlt_down_kernel",nt)], k[("tree_up_kernel",nt)]+k[("qlt_down_kernel",nt)], d["ms_per_step"], d.get("check_ok")))
PY
done
done
