#!/bin/bash
# QLT::run and CAAS::run of a step on two streams: where CAAS starts (bench.py --overlap start | mid | down) against one stream,
# at 1.5M cells, alternating, two passes.
mkdir -p gpurun_out/r04_ovl
for i in 1 2; do
  for m in "--serial" "--overlap start" "--overlap mid" "--overlap down"; do
    timeout -k 10 300 python bench.py $m --no-cpu-baseline --steps 60 > gpurun_out/r04_ovl/b.json 2>> gpurun_out/r04_ovl/err.log || exit 1
    python tests/showbench.py gpurun_out/r04_ovl/b.json | head -1 | sed "s/^/$m: /"
  done
done
