#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p17
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_large.py -x -q > $O/large.log 2>&1; echo "large rc=$?"; tail -3 $O/large.log
CEDR_HIP_APEX=0 timeout -k 10 900 python3 -m pytest tests/test_gpu_large.py -x -q > $O/large_noapex.log 2>&1; echo "large noapex rc=$?"; tail -3 $O/large_noapex.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_large.py -x -q -k "config_C_sharded" > $O/large_c8.log 2>&1; echo "C sharded rc=$?"; tail -3 $O/large_c8.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_large.py -x -q -k "config_D" > $O/large_d8.log 2>&1; echo "D sharded rc=$?"; tail -3 $O/large_d8.log
