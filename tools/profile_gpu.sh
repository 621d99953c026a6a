#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + stats, and SQ counters,
# of a short bench.py run. Outputs under gpurun_out/prof_<tag>/.
set -e
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_${TAG}/kt" -- \
  python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > "gpurun_out/prof_${TAG}_kt.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d "$R/gpurun_out/prof_${TAG}/pmc" -- \
  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "gpurun_out/prof_${TAG}_pmc.log" 2>&1
python3 tools/pmc_summary.py "gpurun_out/prof_${TAG}"
