#!/bin/bash
# Experiment: CAAS's finish pass over the tracers (and chunks) in the reverse of the reduce pass's order: does the memory-side
# cache still hold what the reduce pass touched last?
mkdir -p gpurun_out/r04_finrev
CEDR_HIP_LIB=$PWD/compose_amd/variants/b_finrev.so timeout -k 10 300 python -m pytest tests/test_gpu_caas.py -x -q -m gpu 2>&1 | tail -1
for i in 1 2 3; do
for f in compose_amd/variants/a_base.so compose_amd/variants/b_finrev.so; do
  CEDR_HIP_LIB=$PWD/$f timeout -k 10 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --serial > gpurun_out/r04_finrev/var.json 2> gpurun_out/r04_finrev/var.err || { tail -3 gpurun_out/r04_finrev/var.err; continue; }
  python tests/showbench.py gpurun_out/r04_finrev/var.json | grep "caas_finish\|caas_reduce\|^value" | tr '\n' ' ' | sed "s|^|$(basename $f): |"; echo
done
done
