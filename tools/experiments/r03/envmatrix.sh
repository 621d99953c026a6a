#!/bin/bash
# Round 3: bench.py under environment knobs (one box, alternating): usage envmatrix.sh "<ncells>" "VAR=val ..." ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
mkdir -p gpurun_out/r03env
n=$1; shift
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --ncells $n > gpurun_out/r03env/o.json 2> gpurun_out/r03env/o.err || { tail -3 gpurun_out/r03env/o.err; continue; }
  python3 - "$cfg" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r03env/o.json"))
ks=" ".join("%s/%d=%.4f" % (k["kernel"].replace("_kernel",""), k["launch"], k["avg_ms"]) for k in d["kernels"] if k["kernel"] in ("tree_up_kernel","qlt_down_kernel","qlt_top_kernel","tree_top_up_kernel"))
print("%-34s step %.4f qlt %.4f caas %.4f | %s" % (sys.argv[1], d["ms_per_step"], d["qlt"]["ms_per_run"], d["caas"]["ms_per_run"], ks))
PY
done
done
