#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p8
mkdir -p $O
CEDR_HIP_MICRO_TW=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_qlt.py -x -q > $O/pytest_qlt_tw1.log 2>&1; echo "pytest qlt tw1 rc=$?"; tail -3 $O/pytest_qlt_tw1.log
run () {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --check > $O/$name.json 2> $O/$name.err || { echo "== $name FAILED"; tail -3 $O/$name.err; python3 tests/showbench.py $O/$name.json | head -3; return; }
  echo "== $name"; python3 tests/showbench.py $O/$name.json | grep "qlt_down_kernel *launch \(8192\|1024\|512\)\|value"
}
V=$R/compose_amd/variants
for cut in 2 3; do
  run cut${cut}_tw1_w4spill CEDR_HIP_MICRO_CUT=$cut CEDR_HIP_MICRO_TW=1
  run cut${cut}_tw1_w3 CEDR_HIP_MICRO_CUT=$cut CEDR_HIP_MICRO_TW=1 CEDR_HIP_LIB=$V/sp1_w3.so
  run cut${cut}_tw1_nopf_w4 CEDR_HIP_MICRO_CUT=$cut CEDR_HIP_MICRO_TW=1 CEDR_HIP_LIB=$V/sp1_nopf.so
  run cut${cut}_tw1_nopf_w5 CEDR_HIP_MICRO_CUT=$cut CEDR_HIP_MICRO_TW=1 CEDR_HIP_LIB=$V/sp1_nopf_w5.so
done
