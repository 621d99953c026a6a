#!/bin/bash
# rocprofv3 kernel trace + stats of `bench.py --serial` (QLT and CAAS back to back on
# one stream): the per-kernel averages that bench.py's roofline leg must agree with.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_serial/kt" -- \
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --serial > gpurun_out/prof_serial.json 2> gpurun_out/prof_serial.log
python3 tests/showbench.py gpurun_out/prof_serial.json | head -6
python3 tools/kernel_stats_by_grid.py gpurun_out/prof_serial/kt/*/*kernel_trace.csv gpurun_out/prof_serial_by_grid.csv | head -8
