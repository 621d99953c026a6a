#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p7
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_qlt.py tests/test_gpu_threaded_ranks.py -x -q > $O/pytest_qlt.log 2>&1; echo "pytest qlt rc=$?"; tail -3 $O/pytest_qlt.log
CEDR_HIP_MICRO_TW=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_qlt.py -x -q > $O/pytest_qlt_tw1.log 2>&1; echo "pytest qlt tw1 rc=$?"; tail -3 $O/pytest_qlt_tw1.log
run () {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --check > $O/$name.json 2> $O/$name.err || { echo "== $name FAILED"; tail -3 $O/$name.err; python3 tests/showbench.py $O/$name.json | head -3; return; }
  echo "== $name"; python3 tests/showbench.py $O/$name.json | grep "qlt_down_kernel\|tree_up_kernel\|value"
}
for cut in 1 2 3; do
  run cut${cut}_tw1 CEDR_HIP_MICRO_CUT=$cut CEDR_HIP_MICRO_TW=1
done
run cut2_tw2 CEDR_HIP_MICRO_CUT=2
run cut3_tw2 CEDR_HIP_MICRO_CUT=3
