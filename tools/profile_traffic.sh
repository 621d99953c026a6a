#!/bin/bash
# HBM traffic of the hot kernels from the TCC counters (separate passes, as
# MI355X_MICROARCH.md prescribes). Pass 3/4 repeat with an odd cell count, which
# forces CAAS onto 8-byte accesses: a known-byte-count calibration of FETCH_SIZE for
# the 8-byte access pattern the QLT kernels use.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$R/gpurun_out/prof_traffic/$c" -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial > gpurun_out/prof_traffic_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d "$R/gpurun_out/prof_traffic/odd_$c" -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial --ncells 1500001 > gpurun_out/prof_traffic_odd_$c.log 2>&1
done
python3 tools/traffic_summary.py gpurun_out/prof_traffic
