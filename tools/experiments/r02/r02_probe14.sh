#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p14
mkdir -p $O
run () {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --check > $O/$name.json 2> $O/$name.err || { echo "== $name FAILED"; tail -3 $O/$name.err; grep -q HSA_STATUS $O/$name.err && exit 1; return; }
  echo "== $name"; python3 tests/showbench.py $O/$name.json | grep "tree_up_kernel *launch 8192"
}
V=$R/compose_amd/variants
for pad in 0 8000 14000 22000 36000 60000; do
run pf_pad$pad CEDR_HIP_UP_LDS_PAD=$pad
run nopf_pad$pad CEDR_HIP_UP_LDS_PAD=$pad CEDR_HIP_LIB=$V/up_nopf.so
done
