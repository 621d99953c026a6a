#!/bin/bash
# The leaves->root kernel without its register prefetch (fewer VGPRs: more workgroups per CU) and the small stages' chunks split
# over more workgroups (CEDR_HIP_SPLIT_UP = workgroups wanted): the 1024-tile and 128-tile launches at 1.5M cells and at the shard.
mkdir -p gpurun_out/r04_upnopf
for n in 187500 1500000; do
for f in compose_amd/variants/a_base.so compose_amd/variants/b_upnopf.so; do
  for sp in 1024 2048 3072; do
    CEDR_HIP_SPLIT_UP=$sp CEDR_HIP_LIB=$PWD/$f timeout -k 10 300 python bench.py --ncells $n --steps 40 --warmup 3 --no-cpu-baseline --serial > gpurun_out/r04_upnopf/var.json 2> gpurun_out/r04_upnopf/var.err || { tail -3 gpurun_out/r04_upnopf/var.err; continue; }
    python - "$n $(basename $f) split=$sp" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r04_upnopf/var.json"))
print(sys.argv[1], "qlt %.4f"%d["qlt"]["ms_per_run"], " ".join("%s/%d %.4f"%(k["kernel"][:8],k["launch"],k["avg_ms"]) for k in d["kernels"] if k["kernel"].startswith("tree_up")))
PY
  done
done
done
