#!/bin/bash
# Round 3: the field rows' distance (stride) on physically contiguous buffers, where the address -> channel map is
# the same in every process: first-stage kernels' times by stride.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out/r03env
for cfg in ${CFGS:-"16 0" "16 16" "16 48" "16 112" "16 272" "512 0" "512 16" "512 272" "8192 0" "8192 16" "8192 528" "262144 0" "262144 16" "262144 272" "262144 4112" "262144 33296" "16 0"}; do
  set -- $cfg
  CEDR_HIP_CONTIG=${CONTIG:-1} CEDR_HIP_PLACE_TRIES=1 CEDR_HIP_STRIDE_ALIGN=$1 CEDR_HIP_STRIDE_EXTRA=$2 timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --serial --check > gpurun_out/r03env/sc.json 2> gpurun_out/r03env/sc.err || { tail -3 gpurun_out/r03env/sc.err; continue; }
  python3 - "$cfg" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r03env/sc.json"))
k={ (x["kernel"],x["launch"]):x["avg_ms"] for x in d["kernels"] }
nt=max(b for (a,b) in k if a=="qlt_down_kernel")
print("align,extra %-14s: qlt %.4f  up %.4f down %.4f  sum %.4f check %s" % (sys.argv[1], d["qlt"]["ms_per_run"], k[("tree_up_kernel",nt)], k[("qlt_down_kernel",nt)], k[("tree_up_kernel",nt)]+k[("qlt_down_kernel",nt)], d.get("check_ok")))
PY
done
