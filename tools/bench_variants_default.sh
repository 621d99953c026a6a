#!/bin/bash
# Like bench_variants.sh but the default command (QLT and CAAS on two concurrent streams).
for f in compose_amd/variants/*.so; do
  echo "== $f"
  CEDR_HIP_LIB=$PWD/$f timeout -k 10 300 python bench.py --steps 20 --warmup 3 --check --no-cpu-baseline > gpurun_out/var.json 2> gpurun_out/var.err || { tail -3 gpurun_out/var.err; continue; }
  python tests/showbench.py gpurun_out/var.json | head -5
done
