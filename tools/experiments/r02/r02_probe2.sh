#!/bin/bash
# Round-2 probe 2: parity of the SP down kernel, then variants.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p2
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_qlt.py tests/test_gpu_threaded_ranks.py -x -q > $O/pytest_qlt.log 2>&1; echo "pytest qlt rc=$?"; tail -3 $O/pytest_qlt.log
CEDR_HIP_MICRO_TW=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_qlt.py -x -q > $O/pytest_qlt_tw1.log 2>&1; echo "pytest qlt tw1 rc=$?"; tail -3 $O/pytest_qlt_tw1.log
run () {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --check > $O/$name.json 2> $O/$name.err || { echo "== $name FAILED"; tail -3 $O/$name.err; return; }
  echo "== $name"; python3 tests/showbench.py $O/$name.json | head -4
}
run old CEDR_HIP_DOWN_SP=0
run sp_tw2_w2 X=1
run sp_tw1_w4 CEDR_HIP_MICRO_TW=1
run sp_tw2_w3 CEDR_HIP_LIB=$R/compose_amd/variants/sp_w3.so
run noqp_old CEDR_HIP_LIB=$R/compose_amd/variants/diag_noqp.so CEDR_HIP_DOWN_SP=0
run noqp_sp_tw2 CEDR_HIP_LIB=$R/compose_amd/variants/diag_noqp.so
run noqp_sp_tw1 CEDR_HIP_LIB=$R/compose_amd/variants/diag_noqp.so CEDR_HIP_MICRO_TW=1
