#!/bin/bash
# Cache policy bits on the leaves->root kernel's output stores (p0 = product; 1 sc1, 2 sc0 sc1, 3 nt, 4 sc0 sc1 nt, 5 sc0).
mkdir -p gpurun_out/r04_pol
CEDR_HIP_LIB=$PWD/compose_amd/variants/p2.so timeout -k 10 300 python -m pytest tests/test_gpu_qlt.py -x -q -m gpu 2>&1 | tail -1
for i in 1 2; do
for f in compose_amd/variants/p*.so; do
  CEDR_HIP_LIB=$PWD/$f timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --serial > gpurun_out/r04_pol/var.json 2> gpurun_out/r04_pol/var.err || { tail -3 gpurun_out/r04_pol/var.err; continue; }
  python tests/showbench.py gpurun_out/r04_pol/var.json | grep "tree_up_kernel *launch 8192\|tree_up_kernel *launch 1024\|^value" | tr '\n' ' ' | sed "s|^|$(basename $f): |"; echo
done
done
