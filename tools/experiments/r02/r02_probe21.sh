#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p21
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_mpi.py tests/test_gpu_cpp_dropin.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest mpi+dropin rc=$rc"; tail -12 $O/pytest.log
for cuts in "2 2" "2 0" "3 0" "3 2" "3 3" "4 0" "2 3"; do
  set -- $cuts
  echo "== cut $1 cut2 $2"
  CEDR_HIP_MICRO_CUT=$1 CEDR_HIP_MICRO_CUT2=$2 python3 tools/chain_rank0_of_8.py 2>&1 | grep -v amdgpu.ids | head -2
done
