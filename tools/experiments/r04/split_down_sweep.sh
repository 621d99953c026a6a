#!/bin/bash
# Workgroups wanted for the small root->leaves stages (CEDR_HIP_SPLIT_DOWN; chunks dealt to that many workgroups), with the later
# stages cut at three levels: 1.5M cells and the 187.5k-cell shard, two passes.
mkdir -p gpurun_out/r04_sd
for n in 1500000 187500; do
for i in 1 2; do
  for c in 1024 1280 2048 2560 4096; do
    CEDR_HIP_SPLIT_DOWN=$c timeout -k 10 300 python bench.py --ncells $n --serial --no-cpu-baseline --steps 40 > gpurun_out/r04_sd/b.json 2>> gpurun_out/r04_sd/err.log || { echo "failed"; continue; }
    python - "$n split_down=$c" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r04_sd/b.json"))
print(sys.argv[1], "qlt %.4f ms"%d["qlt"]["ms_per_run"], " ".join("%s/%d %.4f"%(k["kernel"][:8],k["launch"],k["avg_ms"]) for k in d["kernels"] if k["kernel"].startswith(("qlt_down")) and k["launch"]<8192))
PY
  done
done
done
