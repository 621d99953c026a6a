#!/bin/bash
# Round 3: QLT and CAAS on two streams at N = 1, with stream priorities. usage: overlap.sh <ncells>
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
mkdir -p gpurun_out/r03env
n=$1
for rep in 1 2; do
for cfg in ${CFGS:-"serial" "start 0,0" "down 0,0" "mid 0,0" "mid 0,-1" "mid -1,0"}; do
  set -- $cfg
  if [ "$1" = serial ]; then args="--serial"; pr="0,0"; else args="--overlap $1"; pr=$2; fi
  CEDR_BENCH_PRIO=$pr timeout -k 10 200 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --check --ncells $n $args > gpurun_out/r03env/ov.json 2> gpurun_out/r03env/ov.err || { tail -3 gpurun_out/r03env/ov.err; continue; }
  python3 - "$cfg" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r03env/ov.json"))
print("%-14s step %.4f ms  value %.4g  (qlt %.4f caas %.4f alone) check %s" % (sys.argv[1], d["ms_per_step"], d["value"], d["qlt"]["ms_per_run"], d["caas"]["ms_per_run"], d.get("check_ok")))
PY
done
done
