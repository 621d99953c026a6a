#!/bin/bash
# The LDS-DMA build of the root->leaves kernel on the SMALL stages only (their chunk loops are latency bound), at 1.5M
# cells and at the 187.5k-cell shard.
out=gpurun_out/r04_dma_small; mkdir -p $out
for n in 1500000 187500; do
for i in 1 2; do
  for v in "0 0" "1 2047" "1 200"; do
    set -- $v
    CEDR_HIP_DOWN_DMA=$1 CEDR_HIP_DOWN_DMA_MAX=$2 timeout -k 10 300 python bench.py --ncells $n --serial --no-cpu-baseline --steps 40 > $out/b.json 2>> $out/err.log || exit 1
    python - "$1 $2" $n <<'PY'
import json,sys
d=json.load(open("gpurun_out/r04_dma_small/b.json"))
print(sys.argv[2], "dma[%s]"%sys.argv[1], "qlt %.4f ms"%d["qlt"]["ms_per_run"], " ".join("%s/%d %.4f"%(k["kernel"][:8],k["launch"],k["avg_ms"]) for k in d["kernels"] if k["kernel"].startswith(("qlt_down","tree_up","qlt_top"))))
PY
  done
done
done
