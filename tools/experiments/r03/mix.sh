#!/bin/bash
# Round 3: the root->leaves sweep for mixed tracer types (bench.py --types mix): builds x tracers per wave.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
mkdir -p gpurun_out/r03env
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg timeout -k 10 200 python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --types mix > gpurun_out/r03env/m.json 2> gpurun_out/r03env/m.err || { tail -3 gpurun_out/r03env/m.err; continue; }
  python3 - "$cfg" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r03env/m.json"))
ks=" ".join("%s/%d=%.4f" % (k["kernel"].replace("_kernel",""), k["launch"], k["avg_ms"]) for k in d["kernels"] if k["kernel"] in ("tree_up_kernel","qlt_down_kernel","qlt_top_kernel"))
print("%-60s qlt %.4f | %s" % (sys.argv[1], d["qlt"]["ms_per_run"], ks))
PY
done
done
