#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p20
mkdir -p $O
for cuts in "2 2" "2 0" "3 0" "3 2" "3 3" "4 0" "2 3" "2 4"; do
  set -- $cuts
  echo "== cut $1 cut2 $2"
  CEDR_HIP_MICRO_CUT=$1 CEDR_HIP_MICRO_CUT2=$2 python3 tools/chain_rank0_of_8.py 2>&1 | grep -v amdgpu.ids | head -2
done
