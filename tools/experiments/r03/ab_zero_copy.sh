#!/bin/bash
# Round 3: zero-copy QLT::run (bench.py's pack/unpack leg) for the builds in compose_amd/variants. usage: ab_zero_copy.sh v1 v2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out/r03env
for v in "$@"; do
  CEDR_HIP_LIB=$PWD/compose_amd/variants/$v.so timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --serial > gpurun_out/r03env/zc.json 2> gpurun_out/r03env/zc.err || { tail -3 gpurun_out/r03env/zc.err; continue; }
  python3 - "$v" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r03env/zc.json"))
p=d["pack_unpack"]
print("%-16s qlt %.4f  zero-copy run %.4f  set+run+get %.4f" % (sys.argv[1], d["qlt"]["ms_per_run"], p.get("qlt_zero_copy_run_ms", float("nan")), p.get("qlt_set_run_get_ms", float("nan"))), {k: round(v, 4) for k, v in p.items() if isinstance(v, float)})
PY
done
