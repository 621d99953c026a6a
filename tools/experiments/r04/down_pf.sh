#!/bin/bash
# The register-prefetch build of the root->leaves kernel on stages of at most CEDR_HIP_DOWN_PF workgroups: parity (stops at the
# first failure), then timings at 1.5M cells (second and third stage) and at the 187.5k-cell shard (first and second).
out=gpurun_out/r04_pf; mkdir -p $out
CEDR_HIP_DOWN_PF=100000 timeout -k 10 300 python -m pytest tests/test_gpu_qlt.py -x -q -m gpu > $out/pytest1.log 2>&1 || { tail -20 $out/pytest1.log; exit 1; }
tail -1 $out/pytest1.log
CEDR_HIP_DOWN_PF=100000 timeout -k 10 600 python -m pytest tests/test_gpu_threaded_ranks.py tests/test_gpu_large.py -x -q -m gpu > $out/pytest2.log 2>&1 || { tail -20 $out/pytest2.log; exit 1; }
tail -1 $out/pytest2.log
for n in 1500000 187500; do
for i in 1 2 3; do
  for v in 0 300 2100; do
    CEDR_HIP_DOWN_PF=$v timeout -k 10 300 python bench.py --ncells $n --serial --no-cpu-baseline --steps 40 > $out/b.json 2>> $out/err.log || exit 1
    python - "$v" $n <<'PY'
import json,sys
d=json.load(open("gpurun_out/r04_pf/b.json"))
print(sys.argv[2], "pf<=%s"%sys.argv[1], "qlt %.4f ms"%d["qlt"]["ms_per_run"], "step %.4f"%d["ms_per_step"], " ".join("%s/%d %.4f"%(k["kernel"][:8],k["launch"],k["avg_ms"]) for k in d["kernels"] if k["kernel"].startswith(("qlt_down","tree_up","qlt_top"))))
PY
  done
done
done
