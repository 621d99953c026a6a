#!/bin/bash
# Round 5, verdict item 4: three A/Bs on the 14 % of QLT::run that is not first-stage, on today's kernels.
# (i) first stage cut at 3 LDS levels (CEDR_HIP_MICRO_CUT=3) with the later stages at 3 or 4; (ii) two tracers per
# wave in the shape-preserving root->leaves kernel for launches of at most 1024 tiles (CEDR_HIP_SP_TW2_MAX_TILES);
# at 1.5M cells (one rank) and at the 187.5k-cell shard. One box, alternating, two passes. Placement tries as bench.py sets them.
O=gpurun_out/r05_qlt_ab; mkdir -p $O
run () {  # label, n, env...
  local label=$1 n=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --ncells $n --serial --no-cpu-baseline --steps 40 --check > $O/b.json 2>> $O/err.log || { echo "$label failed"; return; }
  python - "$n $label" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r05_qlt_ab/b.json"))
print("%-44s" % sys.argv[1], "qlt %.4f ms" % d["qlt"]["ms_per_run"], "check", d.get("check_ok"), " ".join("%s/%d %.4f"%(k["kernel"][:8],k["launch"],k["avg_ms"]) for k in d["kernels"] if k["kernel"].startswith(("qlt_down","tree_up","qlt_top"))))
PY
}
for n in 1500000 187500; do
for i in 1 2; do
  run "default (cut 2, cut2 3)" $n CEDR_X=0
  run "cut 3, cut2 3" $n CEDR_HIP_MICRO_CUT=3 CEDR_HIP_MICRO_CUT2=3
  run "cut 3, cut2 4" $n CEDR_HIP_MICRO_CUT=3 CEDR_HIP_MICRO_CUT2=4
  run "two tracers per wave up to 1024 tiles" $n CEDR_HIP_SP_TW2_MAX_TILES=1024
  run "two tracers per wave up to 128 tiles" $n CEDR_HIP_SP_TW2_MAX_TILES=128
done
done
