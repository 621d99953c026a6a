#!/bin/bash
# How deep the second (and later) micro stages are cut (CEDR_HIP_MICRO_CUT2 = LDS levels kept; 0 = all): fewer, deeper stages
# against more, shallower ones, at 1.5M cells and at the 187.5k-cell shard.
mkdir -p gpurun_out/r04_cut2
for n in 1500000 187500; do
for i in 1 2; do
  for c in 2 3 4 0; do
    CEDR_HIP_MICRO_CUT2=$c timeout -k 10 300 python bench.py --ncells $n --serial --no-cpu-baseline --steps 40 --check > gpurun_out/r04_cut2/b.json 2>> gpurun_out/r04_cut2/err.log || { echo "cut2=$c failed"; continue; }
    python - "$n cut2=$c" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r04_cut2/b.json"))
print(sys.argv[1], "qlt %.4f ms"%d["qlt"]["ms_per_run"], "check", d.get("check_ok"), " ".join("%s/%d %.4f"%(k["kernel"][:8],k["launch"],k["avg_ms"]) for k in d["kernels"] if k["kernel"].startswith(("qlt_down","tree_up","qlt_top")) and k["launch"]<8192))
PY
  done
done
done
