#!/bin/bash
# Round 3: alternate builds of the library (compose_amd/variants/*.so), bench.py --serial; prints QLT::run and the
# first-stage kernels. usage: ab_variants.sh [passes]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out/r03env
for pass in $(seq 1 ${1:-3}); do
for f in compose_amd/variants/*.so; do
  CEDR_HIP_LIB=$PWD/$f timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --check --no-cpu-baseline --serial ${ARGS} > gpurun_out/r03env/ab.json 2> gpurun_out/r03env/ab.err || { tail -3 gpurun_out/r03env/ab.err; continue; }
  python3 - "$f" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r03env/ab.json"))
k={ (x["kernel"],x["launch"]):x["avg_ms"] for x in d["kernels"] }
def g(n,t):
    for (a,b),v in k.items():
        if a==n and b==t: return "%.4f"%v
    return "?"
nt=max([b for (a,b) in k if a=="qlt_down_kernel"] or [0])
print("%-34s qlt %.4f caas %.4f  down/%d %s up/%d %s  check %s" % (sys.argv[1].split("/")[-1], d["qlt"]["ms_per_run"], d["caas"]["ms_per_run"], nt, g("qlt_down_kernel",nt), nt, g("tree_up_kernel",nt), d.get("check_ok")))
PY
done
done
