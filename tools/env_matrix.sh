#!/bin/bash
# The GPU parity suites under the library's non-default code paths (environment knobs), the fused top
# tile's among them: off, tiny and odd input limits (top tiles at every size, stages above them), with and without
# the apex and micro tiles.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r05envm
mkdir -p $O
echo "# HEAD $(cat .head_sha 2>/dev/null | tr '\n' ' ') kernels_sha256 $(python3 -c 'import compose_amd as c; print(c.kernels_sha256())' 2>/dev/null)"
i=0
for e in "CEDR_HIP_TOP=0" "CEDR_HIP_TOP_MAX=2" "CEDR_HIP_TOP_MAX=37" "CEDR_HIP_TOP_MAX=300" "CEDR_HIP_TOP=0 CEDR_HIP_APEX=0" "CEDR_HIP_APEX=0" \
         "CEDR_HIP_DOWN_SP=0" "CEDR_HIP_MICRO=0" "CEDR_HIP_MICRO=0 CEDR_HIP_TOP_MAX=64" "CEDR_HIP_MICRO=1" "CEDR_HIP_MICRO_CUT=0 CEDR_HIP_MICRO_CUT2=0" \
         "CEDR_HIP_MICRO_CUT=3 CEDR_HIP_MICRO_CUT2=3" "CEDR_HIP_MICRO_CUT=1" "CEDR_HIP_TILE_LEAVES=16" "CEDR_HIP_MICRO_TW=1" "CEDR_HIP_EXCHANGE=2" \
         "CEDR_HIP_EXCHANGE=2 CEDR_HIP_TOP_MAX=16" "CEDR_HIP_TAIL=0" "CEDR_HIP_TAIL=3 CEDR_HIP_TAIL_ROUNDS=1" "CEDR_HIP_MICRO_CUT2=2" "CEDR_HIP_SPLIT_DOWN=2048 CEDR_HIP_SPLIT_UP=2048" \
         "CEDR_HIP_CAAS_BFB_FUSE=0" "CEDR_HIP_CAAS_BFB_P2=0" "CEDR_HIP_BFB_CUT=2" "CEDR_HIP_MICRO_CUT=2 CEDR_HIP_MICRO_CUT2=3" "CEDR_HIP_CAAS_BFB_WGS=3000" \
         "CEDR_HIP_SP_TW2_MAX_TILES=1024" "CEDR_HIP_PLACE_TRIES=4" "CEDR_HIP_DOWN_CLS_MIN_TILES=1" "CEDR_HIP_DOWN_CLS=0" \
         "CEDR_HIP_LEAF_BOUNDS=1 CEDR_HIP_DOWN_CLS_MIN_TILES=64"; do
  i=$((i+1))
  if [ $i -lt ${ENVM_FROM:-1} ] || [ $i -gt ${ENVM_TO:-99} ]; then continue; fi
  if env $e timeout -k 10 600 python3 -m pytest tests/test_gpu_qlt.py tests/test_gpu_threaded_ranks.py tests/test_gpu_bfb.py tests/test_gpu_transport1d.py tests/test_gpu_caas.py tests/test_gpu_large.py -x -q -m gpu > $O/run$i.log 2>&1; then echo "OK   $e: $(tail -1 $O/run$i.log)"; else echo "FAIL $e: $(tail -3 $O/run$i.log | tr '\n' ' ')"; fi
done
