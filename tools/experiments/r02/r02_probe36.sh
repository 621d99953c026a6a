#!/bin/bash
# Leaves->root sweep: tile-major output stores into their own 169 MB buffer (blockstore2) against the earlier
# diagnostic that wrapped them into a 13 MB window (blockstore), no stores at all, and the product.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p36
mkdir -p $O
V=$R/compose_amd/variants
for v in base up_blockstore2 up_blockstore up_nostore base up_blockstore2 up_blockstore up_nostore; do
  if [ $v = base ]; then unset CEDR_HIP_LIB; else export CEDR_HIP_LIB=$V/$v.so; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/$v.json 2> $O/b.err && { echo "== $v"; python3 tests/showbench.py $O/$v.json | grep "tree_up_kernel *launch 8192"; }
done
