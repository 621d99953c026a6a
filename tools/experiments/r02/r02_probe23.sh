#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p23
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_qlt.py tests/test_gpu_bfb.py tests/test_gpu_threaded_ranks.py tests/test_gpu_large.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit 1
for i in 1 2; do
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --check > $O/b$i.json 2> $O/b.err && python3 tests/showbench.py $O/b$i.json | head -4
done
