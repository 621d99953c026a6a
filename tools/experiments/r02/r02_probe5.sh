#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p5
mkdir -p $O
run () {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --check > $O/$name.json 2> $O/$name.err || { echo "== $name FAILED"; tail -3 $O/$name.err; return; }
  echo "== $name"; python3 tests/showbench.py $O/$name.json | grep "qlt_down_kernel\|tree_up_kernel\|value"
}
for cut in 1 2 3 4; do
  run cut${cut}_tw1 CEDR_HIP_MICRO_CUT=$cut CEDR_HIP_MICRO_TW=1
done
run cut2_tw2 CEDR_HIP_MICRO_CUT=2
run cut2_old CEDR_HIP_MICRO_CUT=2 CEDR_HIP_DOWN_SP=0
