#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p13
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_qlt.py tests/test_gpu_bfb.py tests/test_gpu_threaded_ranks.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit 1
run () {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --check > $O/$name.json 2> $O/$name.err || { echo "== $name FAILED"; tail -3 $O/$name.err; grep -q HSA_STATUS $O/$name.err && exit 1; return; }
  echo "== $name"; python3 tests/showbench.py $O/$name.json | grep "tree_up_kernel\|value"
}
V=$R/compose_amd/variants
run ft16 X=1
run ft8 CEDR_HIP_LIB=$V/up_ft8.so
run ft32 CEDR_HIP_LIB=$V/up_ft32.so
run ft16_cut3 CEDR_HIP_MICRO_CUT=3
