#!/bin/bash
# Diagnostic builds (wrong results, timing only): what the leaves->root kernel's output stores cost and where.
# Also the bench data with other perturbations (CEDR_BENCH_PERTURB) on the product build.
mkdir -p gpurun_out/r04_upst
for i in 1 2; do
for f in compose_amd/variants/*.so; do
  CEDR_HIP_LIB=$PWD/$f timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --serial > gpurun_out/r04_upst/var.json 2> gpurun_out/r04_upst/var.err || { tail -3 gpurun_out/r04_upst/var.err; continue; }
  python tests/showbench.py gpurun_out/r04_upst/var.json | grep "tree_up_kernel *launch 8192" | sed "s|^|$(basename $f): |"
done
done
for m in permute small none; do
  CEDR_BENCH_PERTURB=$m timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --serial > gpurun_out/r04_upst/var.json 2> gpurun_out/r04_upst/var.err
  python tests/showbench.py gpurun_out/r04_upst/var.json | grep "qlt_down_kernel *launch 8192\|^value" | sed "s|^|perturb=$m: |"
done
