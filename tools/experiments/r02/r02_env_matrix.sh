#!/bin/bash
# The GPU parity suites under the library's non-default code paths (environment knobs).
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02env
mkdir -p $O
i=0
for e in "CEDR_HIP_APEX=0" "CEDR_HIP_DOWN_SP=0" "CEDR_HIP_MICRO=0" "CEDR_HIP_MICRO=1" "CEDR_HIP_MICRO_CUT=0 CEDR_HIP_MICRO_CUT2=0" "CEDR_HIP_MICRO_CUT=3 CEDR_HIP_MICRO_CUT2=3" "CEDR_HIP_MICRO_CUT=1" "CEDR_HIP_TILE_LEAVES=16" "CEDR_HIP_MICRO_TW=1" "CEDR_HIP_EXCHANGE=2"; do
  i=$((i+1))
  if env $e timeout -k 10 400 python3 -m pytest tests/test_gpu_qlt.py tests/test_gpu_threaded_ranks.py tests/test_gpu_bfb.py tests/test_gpu_transport1d.py -x -q -m gpu > $O/run$i.log 2>&1; then echo "OK   $e: $(tail -1 $O/run$i.log)"; else echo "FAIL $e: $(tail -3 $O/run$i.log | tr '\n' ' ')"; fi
done
