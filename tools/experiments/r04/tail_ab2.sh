#!/bin/bash
# Second A/B of the tail split: k = 0 (off), 2, 3, 5, alternating, four passes, one box; full GPU suite first with the default.
out=gpurun_out/r04_tail2; mkdir -p $out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1 || { tail -20 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
for i in 1 2 3 4; do
  for k in 0 2 3 5; do
    CEDR_HIP_TAIL=$k timeout -k 10 300 python bench.py --serial --no-cpu-baseline --steps 40 > $out/b.json 2>> $out/err.log || exit 1
    python - $k <<'PY'
import json,sys
d=json.load(open("gpurun_out/r04_tail2/b.json"))
print("tail k=%s"%sys.argv[1], "qlt %.4f ms"%d["qlt"]["ms_per_run"], " ".join("%s/%d %.4f"%(k["kernel"][:8],k["launch"],k["avg_ms"]) for k in d["kernels"] if k["launch"]==8192))
PY
  done
done
