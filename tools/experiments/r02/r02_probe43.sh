#!/bin/bash
# Root->leaves sweep: wave priority by phase. tXlY = priority X from a chunk's loads to its first barrier,
# Y during the LDS levels' node problems, 0 in the register phase.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p43
mkdir -p $O
V=$R/compose_amd/variants
for rep in 1 2 3; do for v in base prio1 prio_t1l1 prio_t2l1 prio_t1l2 prio_t3l3; do
  if [ $v = base ]; then unset CEDR_HIP_LIB; else export CEDR_HIP_LIB=$V/$v.so; fi
  timeout -k 10 300 python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/$v.json 2> $O/b.err && { echo -n "$v "; python3 tests/showbench.py $O/$v.json | grep "qlt_down_kernel *launch 8192"; }
done; done
