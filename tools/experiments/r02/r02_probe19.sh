#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p19
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_multirank.py tests/test_gpu_large.py -x -q > $O/two.log 2>&1; echo "multirank+large rc=$?"; tail -2 $O/two.log
python3 tools/chain_rank0_of_8.py 2>&1 | grep -v amdgpu.ids | tee $O/chain.txt
