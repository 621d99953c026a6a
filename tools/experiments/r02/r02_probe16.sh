#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p16
mkdir -p $O
for f in test_gpu_bfb test_gpu_multirank test_gpu_threaded_ranks; do
  timeout -k 10 600 python3 -m pytest tests/$f.py -x -q > $O/$f.log 2>&1; echo "$f rc=$?"; tail -2 $O/$f.log
  CEDR_HIP_APEX=0 timeout -k 10 600 python3 -m pytest tests/$f.py -x -q > $O/${f}_noapex.log 2>&1; echo "$f noapex rc=$?"; tail -2 $O/${f}_noapex.log
done
timeout -k 10 600 python3 -m pytest tests/test_gpu_threaded_ranks.py -x -q -k "qlt_ranks_as_threads" > $O/t1.log 2>&1; echo "qlt_ranks rc=$?"; tail -2 $O/t1.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_threaded_ranks.py -x -q -k "bfb" > $O/t2.log 2>&1; echo "bfb rc=$?"; tail -2 $O/t2.log
