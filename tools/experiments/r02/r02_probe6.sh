#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p6
mkdir -p $O
for sp in 0 1; do for cut in 1 2; do
CEDR_HIP_DOWN_SP=$sp CEDR_HIP_MICRO_CUT=$cut timeout -k 10 900 python3 -m pytest tests/test_gpu_qlt.py -x -q > $O/pytest_sp${sp}_cut$cut.log 2>&1; echo "pytest sp=$sp cut=$cut rc=$?"; tail -4 $O/pytest_sp${sp}_cut$cut.log
done; done
CEDR_HIP_DOWN_SP=0 CEDR_HIP_MICRO_CUT=1 timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --check > $O/old_cut1.json 2> $O/old_cut1.err; echo "old cut1 bench rc=$?"
