#!/bin/bash
# Shard-sized problems on one GPU (what each of N GPUs runs before the collectives), one stream and
# two streams, and the rank-0-of-8 chain with a stub communicator. Output: profiles/rNN_shards_and_chain.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/${RN:-r04}shards
mkdir -p $O
for n in 1500000 750000 375000 187500; do
  timeout -k 10 200 python3 bench.py --serial --steps 100 --warmup 5 --no-cpu-baseline --check --ncells $n > $O/s_$n.json 2> $O/b.err && { echo -n "$n cells, one stream: "; python3 tests/showbench.py $O/s_$n.json | grep "value"; }
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --check --ncells $n > $O/t_$n.json 2> $O/b.err && { echo -n "$n cells, two streams (default): "; python3 tests/showbench.py $O/t_$n.json | grep "value"; }
done
for ov in down mid; do
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --check --ncells 187500 --overlap $ov > $O/s_187500_$ov.json 2> $O/b.err && { echo -n "187500 cells, two streams ($ov): "; python3 tests/showbench.py $O/s_187500_$ov.json | grep "value"; }
done
timeout -k 10 200 python3 tools/chain_rank0_of_8.py 2>&1 | grep -v amdgpu.ids | tail -44
