#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p25
mkdir -p $O
for v in base l2 base l2; do
  if [ $v = l2 ]; then export CEDR_HIP_LIB=$R/compose_amd/variants/diag_l2.so; else unset CEDR_HIP_LIB; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/$v.json 2> $O/b.err && { echo "== $v"; python3 tests/showbench.py $O/$v.json | grep "qlt_down_kernel *launch 8192\|tree_up_kernel *launch 8192"; }
done
