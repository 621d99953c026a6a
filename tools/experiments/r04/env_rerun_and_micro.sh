#!/bin/bash
for e in "CEDR_HIP_EXCHANGE=2" "CEDR_HIP_EXCHANGE=2 CEDR_HIP_TOP_MAX=16"; do
  if env $e timeout -k 10 600 python3 -m pytest tests/test_gpu_qlt.py tests/test_gpu_threaded_ranks.py tests/test_gpu_bfb.py tests/test_gpu_transport1d.py tests/test_gpu_caas.py tests/test_gpu_large.py -x -q -m gpu > gpurun_out/r04envm_rerun.log 2>&1; then echo "OK   $e: $(tail -1 gpurun_out/r04envm_rerun.log)"; else echo "FAIL $e: $(tail -3 gpurun_out/r04envm_rerun.log | tr '\n' ' ')"; fi
done
true
