#!/bin/bash
# Experiment: the root->leaves sweep takes every XCD's tiles in DESCENDING order (the leaves->root sweep ascending), so that it starts
# with the cells the leaves->root sweep read last: does the 256 MB memory-side cache still hold some of them?
out=gpurun_out/r04_rev; mkdir -p $out
CEDR_HIP_DOWN_REVERSE=1 timeout -k 10 600 python -m pytest tests/test_gpu_qlt.py tests/test_gpu_large.py -x -q -m gpu > $out/pytest.log 2>&1 || { tail -20 $out/pytest.log; exit 1; }
tail -1 $out/pytest.log
for i in 1 2 3; do
  for v in 0 1; do
    CEDR_HIP_DOWN_REVERSE=$v timeout -k 10 300 python bench.py --serial --no-cpu-baseline --steps 40 > $out/b.json 2>> $out/err.log || exit 1
    python tests/showbench.py $out/b.json | head -3 | tr '\n' ' ' | sed "s/^/reverse=$v: /"; echo
  done
done
