#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p18
mkdir -p $O
for i in 1 2; do
timeout -k 10 900 python3 -m pytest tests/test_gpu_threaded_ranks.py tests/test_gpu_bfb.py tests/test_gpu_multirank.py tests/test_gpu_large.py -x -q > $O/all$i.log 2>&1; echo "all4 #$i rc=$?"; tail -2 $O/all$i.log
done
timeout -k 10 900 python3 -m pytest tests/test_gpu_threaded_ranks.py tests/test_gpu_bfb.py tests/test_gpu_multirank.py tests/test_gpu_large.py -x -q --deselect tests/test_gpu_multirank.py::test_rccl_single_rank > $O/norccl.log 2>&1; echo "without rccl test rc=$?"; tail -2 $O/norccl.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_multirank.py tests/test_gpu_large.py -x -q > $O/two.log 2>&1; echo "multirank+large rc=$?"; tail -2 $O/two.log
