#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p15
mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests/test_gpu_threaded_ranks.py tests/test_gpu_bfb.py tests/test_gpu_multirank.py tests/test_gpu_large.py -x -q > $O/pytest_apex.log 2>&1; rc=$?; echo "pytest apex rc=$rc"; tail -3 $O/pytest_apex.log
[ $rc -ne 0 ] && { grep -a "Error\|assert\|FAILED" $O/pytest_apex.log | head; exit 1; }
CEDR_HIP_APEX=0 timeout -k 10 900 python3 -m pytest tests/test_gpu_threaded_ranks.py -x -q > $O/pytest_noapex.log 2>&1; rc=$?; echo "pytest no-apex rc=$rc"; tail -3 $O/pytest_noapex.log
python3 tools/chain_rank0_of_8.py 2>&1 | grep -v amdgpu.ids | tee $O/chain.txt
