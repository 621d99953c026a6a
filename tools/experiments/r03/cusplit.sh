#!/bin/bash
# Round 3: QLT::run and CAAS::run on CU-masked streams (bench.py --cu-split). usage: cusplit.sh <ncells>
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
mkdir -p gpurun_out/r03env
n=$1
for rep in 1 2; do
for cfg in ${CFGS:-"serial 0" "start 0" "start 1" "start 2" "down 1" "down 2" "mid 1" "mid 2" "start 3"}; do
  set -- $cfg
  if [ "$1" = serial ]; then args="--serial"; else args="--overlap $1 --cu-split $2"; fi
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --check --ncells $n $args > gpurun_out/r03env/cs.json 2> gpurun_out/r03env/cs.err || { tail -3 gpurun_out/r03env/cs.err; continue; }
  python3 - "$cfg" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r03env/cs.json"))
print("%-10s step %.4f ms  value %.4g  (qlt %.4f caas %.4f alone) check %s" % (sys.argv[1], d["ms_per_step"], d["value"], d["qlt"]["ms_per_run"], d["caas"]["ms_per_run"], d.get("check_ok")))
PY
done
done
