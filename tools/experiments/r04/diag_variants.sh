#!/bin/bash
# Diagnostic builds of the library (wrong results, timing only): where the first-stage root->leaves kernel's time is.
# With and without the LDS-DMA prefetch (CEDR_HIP_DOWN_DMA).
mkdir -p gpurun_out/r04_diag
for f in compose_amd/variants/*.so; do
  for v in 0 2048; do
    CEDR_HIP_DOWN_DMA=$v CEDR_HIP_LIB=$PWD/$f timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --serial > gpurun_out/r04_diag/var.json 2> gpurun_out/r04_diag/var.err || { tail -3 gpurun_out/r04_diag/var.err; continue; }
    python tests/showbench.py gpurun_out/r04_diag/var.json | grep "launch 8192" | sed "s|^|$(basename $f) dma>=$v: |"
  done
done
