#!/bin/bash
# CAAS kernels of several builds (compose_amd/variants/*.so), alternating, twice.
for rep in 1 2; do
for f in compose_amd/variants/*.so; do
  CEDR_HIP_LIB=$PWD/$f timeout -k 10 300 python bench.py --steps 30 --warmup 3 --check --no-cpu-baseline > gpurun_out/var.json 2> gpurun_out/var.err || { tail -3 gpurun_out/var.err; continue; }
  python3 - "$f" <<'PY'
import json,sys
d=json.load(open("gpurun_out/var.json"))
k={x["kernel"]:x["avg_ms"] for x in d["kernels"] if x["kernel"].startswith("caas")}
print("%-44s caas %.4f ms | reduce %.4f finish %.4f sum %.4f  check %s" % (sys.argv[1].split("/")[-1], d["caas"]["ms_per_run"], k.get("caas_reduce_kernel",0), k.get("caas_finish_kernel",0), k.get("caas_sum_partials_kernel",0), d.get("check_ok")))
PY
done
done
