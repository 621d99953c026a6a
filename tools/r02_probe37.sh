#!/bin/bash
# Blocked (tile-major) first-stage outputs, product build: CEDR_HIP_BLOCKED=0 against the default, same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p37
mkdir -p $O
for rep in 1 2; do for b in 0 1; do
  export CEDR_HIP_BLOCKED=$b
  timeout -k 10 300 python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --check > $O/b$b.json 2> $O/b.err && { echo "== blocked $b"; python3 tests/showbench.py $O/b$b.json | grep "value\|launch 8192\|launch 1024"; }
done; done
for n in 187500; do for b in 0 1; do
  export CEDR_HIP_BLOCKED=$b
  timeout -k 10 300 python3 bench.py --steps 100 --warmup 3 --no-cpu-baseline --check --ncells $n > $O/b${b}_$n.json 2> $O/b.err && { echo "== blocked $b, $n cells"; python3 tests/showbench.py $O/b${b}_$n.json | grep "value\|launch 1024\|launch 128 "; }
done; done
