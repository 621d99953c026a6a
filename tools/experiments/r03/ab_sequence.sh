#!/bin/bash
# Round 3: does a build's leaves->root time follow the build or the order of the processes? Runs the variants in the
# order given (names without .so). usage: ab_sequence.sh a_base a_base b_new b_new ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out/r03env
for v in "$@"; do
  f=compose_amd/variants/$v.so
  CEDR_HIP_LIB=$PWD/$f timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --serial > gpurun_out/r03env/ab.json 2> gpurun_out/r03env/ab.err || { tail -3 gpurun_out/r03env/ab.err; continue; }
  python3 - "$v" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r03env/ab.json"))
k={ (x["kernel"],x["launch"]):x["avg_ms"] for x in d["kernels"] }
nt=max(b for (a,b) in k if a=="qlt_down_kernel")
print("%-20s qlt %.4f caas %.4f  down/%d %.4f up/%d %.4f" % (sys.argv[1], d["qlt"]["ms_per_run"], d["caas"]["ms_per_run"], nt, k[("qlt_down_kernel",nt)], nt, k[("tree_up_kernel",nt)]))
PY
done
