#!/bin/bash
# A/B of the root->leaves kernel's LDS-DMA build (CEDR_HIP_DOWN_DMA = minimum tiles of a stage; 0 = off) on one box:
# parity first, then alternating bench.py --serial runs.
out=gpurun_out/r04_dma; mkdir -p $out
CEDR_HIP_DOWN_DMA=1 timeout -k 10 600 python -m pytest tests/test_gpu_qlt.py tests/test_gpu_large.py -x -q -m gpu > $out/pytest_dma.log 2>&1 || { tail -20 $out/pytest_dma.log; exit 1; }
tail -2 $out/pytest_dma.log
for i in 1 2 3; do
  for v in 0 2048; do
    CEDR_HIP_DOWN_DMA=$v timeout -k 10 300 python bench.py --serial --no-cpu-baseline --steps 40 > $out/bench_${v}_$i.json 2>> $out/err.log || exit 1
    python tests/showbench.py $out/bench_${v}_$i.json | head -4 | sed "s/^/dma>=$v: /"
  done
done
