#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p12
mkdir -p $O
run () {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --check > $O/$name.json 2> $O/$name.err || { echo "== $name FAILED"; tail -3 $O/$name.err; grep -q HSA_STATUS $O/$name.err && exit 1; return; }
  echo "== $name"; python3 tests/showbench.py $O/$name.json | grep "tree_up_kernel\|value"
}
V=$R/compose_amd/variants
run wait_first X=1
run wait_last CEDR_HIP_LIB=$V/up_wait_last.so
run ft32 CEDR_HIP_LIB=$V/up_ft32.so
run ft8 CEDR_HIP_LIB=$V/up_ft8.so
run wait_first_cut3 CEDR_HIP_MICRO_CUT=3
