#!/bin/bash
# Round 3: the first-stage kernels over QLT's own l2r buffer, first allocation against best of 4 (CEDR_HIP_PLACE_TRIES).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out/r03env
for pass in 1 2 3 4 5; do
for t in 1 4; do
  CEDR_HIP_PLACE_TRIES=$t timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --serial ${ARGS} > gpurun_out/r03env/pt.json 2> gpurun_out/r03env/pt.err || { tail -3 gpurun_out/r03env/pt.err; continue; }
  python3 - "$t" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r03env/pt.json"))
k={ (x["kernel"],x["launch"]):x["avg_ms"] for x in d["kernels"] }
nt=max(b for (a,b) in k if a=="qlt_down_kernel")
print("tries %s: qlt %.4f  up/%d %.4f down/%d %.4f  step %.4f" % (sys.argv[1], d["qlt"]["ms_per_run"], nt, k[("tree_up_kernel",nt)], nt, k[("qlt_down_kernel",nt)], d["ms_per_step"]))
PY
done
done
