#!/bin/bash
# Shard-sized problems on one GPU (what each of N GPUs runs before the collectives), one stream and
# two streams, and the rank-0-of-8 chain with a stub communicator. Output: profiles/rNN_shards_and_chain.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/${RN:-r05}shards
mkdir -p $O
for n in 1500000 750000 375000 187500; do
  timeout -k 10 200 python3 bench.py --serial --steps 100 --warmup 5 --no-cpu-baseline --check --ncells $n > $O/s_$n.json 2> $O/b.err && { echo -n "$n cells, one stream: "; python3 tests/showbench.py $O/s_$n.json | grep "value"; }
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --check --ncells $n > $O/t_$n.json 2> $O/b.err && { echo -n "$n cells, two streams (default): "; python3 tests/showbench.py $O/t_$n.json | grep "value"; }
done
for ov in down mid; do
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --check --ncells 187500 --overlap $ov > $O/s_187500_$ov.json 2> $O/b.err && { echo -n "187500 cells, two streams ($ov): "; python3 tests/showbench.py $O/s_187500_$ov.json | grep "value"; }
done
# Rank 0 of 8 on this one GPU (bench.py --pretend-ranks 8: the 187.5k-cell shard WITH QLT::run's all-gather + apex and CAAS::run's
# all-reduce; the other ranks' all-gather blocks are copies of this rank's): over the single-rank communicator (collectives =
# stream-ordered copies) and over an RCCL communicator of one rank (--force-rccl: librccl's own enqueue path + the library's
# cross-stream ordering events). The difference is what the real library costs a step per collective pair, before any wire time.
for rc in "" "--force-rccl"; do
  for st in "" "--serial"; do
    for bf in "" "--caas-bfb 4"; do
      timeout -k 10 200 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --check --pretend-ranks 8 $rc $st $bf > $O/r.json 2> $O/b.err && {
        echo -n "rank 0 of 8 (rehearsal), ${rc:-self communicator}, ${st:-two streams}, ${bf:-own all-reduce}: "; python3 - $O/r.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("step %.4f ms  QLT::run %.4f ms  CAAS::run %.4f ms  check %s" % (d["ms_per_step"], d["qlt"]["ms_per_run"], d["caas"]["ms_per_run"], d.get("check_ok")))
PY
      } || tail -2 $O/b.err
    done
  done
done
timeout -k 10 200 python3 tools/chain_rank0_of_8.py 2>&1 | grep -v amdgpu.ids | tail -44
