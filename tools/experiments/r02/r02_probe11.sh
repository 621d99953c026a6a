#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p11
mkdir -p $O
timeout -k 5 120 tools/bin/micro_stream | tee $O/micro_stream.txt
run () {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/$name.json 2> $O/$name.err || { echo "== $name FAILED"; tail -3 $O/$name.err; grep -q HSA_STATUS $O/$name.err && exit 1; return; }
  echo "== $name"; python3 tests/showbench.py $O/$name.json | grep "qlt_down_kernel\|tree_up_kernel\|value"
}
run default X=1
run l2 CEDR_HIP_LIB=$R/compose_amd/variants/diag_l2.so
