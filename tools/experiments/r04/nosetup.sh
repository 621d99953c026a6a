#!/bin/bash
# Diagnostic builds (wrong results): the root->leaves kernel's skeleton (no node problems) with and without the per-tile setup
# (density loads, divisions, level sums).
mkdir -p gpurun_out/r04_nosetup
for i in 1 2 3; do
for f in compose_amd/variants/*.so; do
  CEDR_HIP_LIB=$PWD/$f timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --serial > gpurun_out/r04_nosetup/var.json 2> gpurun_out/r04_nosetup/var.err || { tail -3 gpurun_out/r04_nosetup/var.err; continue; }
  python tests/showbench.py gpurun_out/r04_nosetup/var.json | grep "qlt_down_kernel *launch 8192\|qlt_down_kernel *launch 1024" | sed "s|^|$(basename $f): |"
done
done
