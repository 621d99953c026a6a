#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p10
mkdir -p $O
CEDR_HIP_MICRO_CUT=2 CEDR_HIP_MICRO_CUT2=3 timeout -k 10 900 python3 -m pytest tests/test_gpu_qlt.py tests/test_gpu_threaded_ranks.py -x -q > $O/pytest_qlt.log 2>&1; rc=$?; echo "pytest qlt cut 2,3 rc=$rc"; tail -3 $O/pytest_qlt.log
[ $rc -ne 0 ] && exit 1
run () {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --check > $O/$name.json 2> $O/$name.err || { echo "== $name FAILED"; tail -3 $O/$name.err; python3 tests/showbench.py $O/$name.json | head -3; grep -q HSA_STATUS $O/$name.err && exit 1; return; }
  echo "== $name"; python3 tests/showbench.py $O/$name.json | grep "qlt_down_kernel\|tree_up_kernel\|value"
}
for cuts in "2 0" "2 2" "2 3" "2 4" "3 0" "3 2" "3 3"; do
  set -- $cuts
  run cut$1_$2 CEDR_HIP_MICRO_CUT=$1 CEDR_HIP_MICRO_CUT2=$2
done
