#!/bin/bash
# Root->leaves sweep: s_setprio around the LDS levels' node problems (the phase in which one or two waves of a
# workgroup work and the others wait at the barrier), same box, alternating.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p42
mkdir -p $O
V=$R/compose_amd/variants
for rep in 1 2 3 4; do for v in base prio3 prio2 prio1; do
  if [ $v = base ]; then unset CEDR_HIP_LIB; else export CEDR_HIP_LIB=$V/$v.so; fi
  timeout -k 10 300 python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/$v.json 2> $O/b.err && { echo -n "$v "; python3 tests/showbench.py $O/$v.json | grep "qlt_down_kernel *launch 8192"; }
done; done
