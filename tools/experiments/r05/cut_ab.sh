#!/bin/bash
# The first-stage cut again, on whatever box this call gets (boxes differ in the kind of memory hipMalloc hands out):
# default (2 LDS levels, later stages 3) against 3 / 3 and 3 / 4, two-stream step included. Three alternating passes.
O=gpurun_out/r05_qlt_ab; mkdir -p $O
run () {
  local label=$1 n=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --ncells $n --no-cpu-baseline --steps 40 --check > $O/c.json 2>> $O/err.log || { echo "$label failed"; return; }
  python - "$n $label" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r05_qlt_ab/c.json"))
print("%-34s" % sys.argv[1], "step %.4f serial %.4f qlt %.4f ms" % (d["ms_per_step"], d["serial"]["ms_per_step"], d["qlt"]["ms_per_run"]), "check", d.get("check_ok"), " ".join("%s/%d %.4f"%(k["kernel"][:8],k["launch"],k["avg_ms"]) for k in d["kernels"] if k["kernel"].startswith(("qlt_down","tree_up","qlt_top")) and k["launch"] >= 512), d["placement"]["report"])
PY
}
for i in 1 2 3; do
  for n in 1500000 187500; do
  run "default (cut 2, cut2 3)" $n CEDR_X=0
  run "cut 3, cut2 3" $n CEDR_HIP_MICRO_CUT=3 CEDR_HIP_MICRO_CUT2=3
  run "cut 3, cut2 4" $n CEDR_HIP_MICRO_CUT=3 CEDR_HIP_MICRO_CUT2=4
  done
done
