#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p9
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_qlt.py tests/test_gpu_threaded_ranks.py tests/test_gpu_large.py -x -q > $O/pytest_qlt.log 2>&1; rc=$?; echo "pytest qlt rc=$rc"; tail -3 $O/pytest_qlt.log
[ $rc -ne 0 ] && exit 1
CEDR_HIP_MICRO_TW=1 CEDR_HIP_MICRO_CUT=2 timeout -k 10 900 python3 -m pytest tests/test_gpu_qlt.py -x -q > $O/pytest_qlt_tw1.log 2>&1; rc=$?; echo "pytest qlt tw1 cut2 rc=$rc"; tail -3 $O/pytest_qlt_tw1.log
[ $rc -ne 0 ] && exit 1
run () {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --check > $O/$name.json 2> $O/$name.err || { echo "== $name FAILED"; tail -3 $O/$name.err; python3 tests/showbench.py $O/$name.json | head -3; grep -q HSA_STATUS $O/$name.err && exit 1; return; }
  echo "== $name"; python3 tests/showbench.py $O/$name.json | grep "qlt_down_kernel *launch \(8192\|1024\|512\)\|value"
}
V=$R/compose_amd/variants
for cut in 2 3; do
  run cut${cut}_tw1_w5 CEDR_HIP_MICRO_CUT=$cut CEDR_HIP_MICRO_TW=1
  run cut${cut}_tw1_w4 CEDR_HIP_MICRO_CUT=$cut CEDR_HIP_MICRO_TW=1 CEDR_HIP_LIB=$V/sp1_w4.so
  run cut${cut}_tw2_w3 CEDR_HIP_MICRO_CUT=$cut
done
