#!/bin/bash
# Run bench.py against several builds of the library (compose_amd/variants/*.so).
for f in compose_amd/variants/*.so; do
  echo "== $f"
  CEDR_HIP_LIB=$PWD/$f timeout -k 10 300 python bench.py --steps 10 --warmup 2 --check --no-cpu-baseline --serial > gpurun_out/var.json 2> gpurun_out/var.err || { tail -3 gpurun_out/var.err; continue; }
  python tests/showbench.py gpurun_out/var.json | head -4
done
