#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out/r03env
for pass in 1 2 3; do
for t in 4 8; do
  CEDR_HIP_PLACE_VERBOSE=1 CEDR_HIP_PLACE_TRIES=$t timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --serial > gpurun_out/r03env/pt.json 2> gpurun_out/r03env/pt.err || { tail -3 gpurun_out/r03env/pt.err; continue; }
  grep "placement try" gpurun_out/r03env/pt.err | sed 's/cedr: l2r placement //' | tr '\n' ';'; echo
  python3 - "$t" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r03env/pt.json"))
k={ (x["kernel"],x["launch"]):x["avg_ms"] for x in d["kernels"] }
nt=max(b for (a,b) in k if a=="qlt_down_kernel")
print("tries %s: qlt %.4f  up/%d %.4f down/%d %.4f  sum %.4f step %.4f" % (sys.argv[1], d["qlt"]["ms_per_run"], nt, k[("tree_up_kernel",nt)], nt, k[("qlt_down_kernel",nt)], k[("tree_up_kernel",nt)]+k[("qlt_down_kernel",nt)], d["ms_per_step"]))
PY
done
done
