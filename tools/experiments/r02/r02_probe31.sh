#!/bin/bash
# QLT::run and CAAS::run of one step on two streams against one stream, full size:
# --overlap start (both begin together), --overlap down (CAAS begins with QLT's first-stage root->leaves sweep).
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p31
mkdir -p $O
for v in serial start down serial start down; do
  if [ $v = serial ]; then f=; else f="--overlap $v"; fi
  timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --check $f > $O/$v.json 2> $O/b.err && { echo "== $v"; python3 tests/showbench.py $O/$v.json | grep "value"; }
done
for n in 750000 187500; do for v in serial start down; do
  if [ $v = serial ]; then f=; else f="--overlap $v"; fi
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --check --ncells $n $f > $O/${v}_$n.json 2> $O/b.err && { echo "== $v $n"; python3 tests/showbench.py $O/${v}_$n.json | grep "value"; }
done; done
