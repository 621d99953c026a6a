#!/bin/bash
# Wave priorities: the product (root->leaves sweep with priorities by phase) against no priorities, and the
# leaves->root sweep with priority on its load issue; same box, alternating.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p44
mkdir -p $O
V=$R/compose_amd/variants
for rep in 1 2 3; do for v in base noprio upprio; do
  if [ $v = base ]; then unset CEDR_HIP_LIB; else export CEDR_HIP_LIB=$V/$v.so; fi
  timeout -k 10 300 python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --check > $O/$v.json 2> $O/b.err && { echo "== $v"; python3 tests/showbench.py $O/$v.json | grep "value\|launch 8192\|launch 1024"; }
done; done
