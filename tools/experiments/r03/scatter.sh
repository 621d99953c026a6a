#!/bin/bash
# Round 3: QLT's l2r (and r2l) buffer as 2 MB physical chunks mapped in shuffled order (CEDR_HIP_SCATTER) against hipMalloc.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out/r03env
for pass in 1 2 3; do
for cfg in ${CFGS:-"0 2097152" "1 2097152" "2 2097152" "2 65536" "2 33554432"}; do
  set -- $cfg
  CEDR_HIP_SCATTER=$1 CEDR_HIP_SCATTER_CHUNK=$2 CEDR_HIP_PLACE_TRIES=1 timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --serial --check > gpurun_out/r03env/st.json 2> gpurun_out/r03env/st.err || { tail -3 gpurun_out/r03env/st.err; continue; }
  python3 - "$cfg" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r03env/st.json"))
k={ (x["kernel"],x["launch"]):x["avg_ms"] for x in d["kernels"] }
nt=max(b for (a,b) in k if a=="qlt_down_kernel")
print("scatter,chunk %-12s: qlt %.4f  up %.4f down %.4f  sum %.4f check %s" % (sys.argv[1], d["qlt"]["ms_per_run"], k[("tree_up_kernel",nt)], k[("qlt_down_kernel",nt)], k[("tree_up_kernel",nt)]+k[("qlt_down_kernel",nt)], d.get("check_ok")))
PY
done
done
