#!/bin/bash
# Round 3: chunk splits of the first-stage kernels at the 187.5k-cell shard (rank 0 of 8, stub communicator).
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
for e in "CEDR_X=0" "CEDR_HIP_SPLIT_UP=1536" "CEDR_HIP_SPLIT_UP=2048" "CEDR_HIP_SPLIT_UP=3072" "CEDR_HIP_SPLIT_UP=4096" "CEDR_HIP_SPLIT_DOWN=1024" "CEDR_HIP_SPLIT_DOWN=3072" "CEDR_HIP_SPLIT_DOWN=5120" "CEDR_X=0"; do
  echo "== $e"
  env $e timeout -k 10 200 python3 tools/chain_rank0_of_8.py 2>&1 | grep -v amdgpu.ids | sed -n '2p;13,19p'
done
