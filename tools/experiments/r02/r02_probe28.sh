#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p28
mkdir -p $O
V=$R/compose_amd/variants
for v in base defer base defer; do
  if [ $v = base ]; then unset CEDR_HIP_LIB; export CEDR_HIP_UP_LDS_PAD=0; else export CEDR_HIP_LIB=$V/up_defer.so; export CEDR_HIP_UP_LDS_PAD=21000; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --check > $O/$v.json 2> $O/b.err; echo "== $v rc=$?"; python3 tests/showbench.py $O/$v.json | grep "value\|tree_up_kernel *launch \(8192\|1024\)"
done
