#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p3
mkdir -p $O
run () {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/$name.json 2> $O/$name.err || { echo "== $name FAILED"; tail -3 $O/$name.err; return; }
  echo "== $name"; python3 tests/showbench.py $O/$name.json | grep "qlt_down_kernel *launch 8192\|tree_up_kernel *launch 8192"
}
V=$R/compose_amd/variants
run noqp_old CEDR_HIP_LIB=$V/diag_noqp.so CEDR_HIP_DOWN_SP=0
run noqp_sp_tw2 CEDR_HIP_LIB=$V/diag_noqp.so
run noqp_sp_tw1 CEDR_HIP_LIB=$V/diag_noqp.so CEDR_HIP_MICRO_TW=1
run l2_old CEDR_HIP_LIB=$V/diag_l2.so CEDR_HIP_DOWN_SP=0
run l2_sp_tw2 CEDR_HIP_LIB=$V/diag_l2.so
run l2_sp_tw1 CEDR_HIP_LIB=$V/diag_l2.so CEDR_HIP_MICRO_TW=1
run l2noqp_old CEDR_HIP_LIB=$V/diag_l2_noqp.so CEDR_HIP_DOWN_SP=0
run l2noqp_sp_tw2 CEDR_HIP_LIB=$V/diag_l2_noqp.so
run l2noqp_sp_tw1 CEDR_HIP_LIB=$V/diag_l2_noqp.so CEDR_HIP_MICRO_TW=1
