#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p24
mkdir -p $O
for sp in 3072 9000 17000 3072 9000; do
  CEDR_HIP_SPLIT_DOWN=$sp timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/b$sp.json 2> $O/b.err && { echo "== split $sp"; python3 tests/showbench.py $O/b$sp.json | grep "value\|qlt_down_kernel *launch \(8192\|1024\)"; }
done
