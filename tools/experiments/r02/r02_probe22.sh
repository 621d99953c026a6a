#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p22
mkdir -p $O
for n in 187500 375000 750000; do
for mode in "" "--overlap"; do
  timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --ncells $n $mode > $O/b_${n}_$mode.json 2> $O/b.err || { tail -3 $O/b.err; continue; }
  echo "== ncells $n $mode"; python3 tests/showbench.py $O/b_${n}_$mode.json | head -1
done; done
