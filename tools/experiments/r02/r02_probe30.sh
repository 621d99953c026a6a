#!/bin/bash
# Row stride against DRAM bank / channel placement: the same sweeps with the field rows' stride rounded
# to different moduli (CEDR_HIP_STRIDE_MOD=M,O doubles).
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p30
mkdir -p $O
for rep in 1 2; do
for m in 16,0 512,0 512,16 512,256 4096,16 32768,0 32768,16 32768,2064 262144,0 262144,16 262144,4112; do
  export CEDR_HIP_STRIDE_MOD=$m
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/s_$m.json 2> $O/b.err && { echo "== stride mod $m"; python3 tests/showbench.py $O/s_$m.json | grep "value\|launch 8192" | cut -c1-90; }
done
done
