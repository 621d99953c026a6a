#!/bin/bash
# Round-2 first probe on the GPU box: multirank bench tests, baseline, phase stamps, SQ counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02p1
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_multirank.py -x -q > $O/pytest_multirank.log 2>&1; echo "pytest multirank rc=$?"
tail -3 $O/pytest_multirank.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --check > $O/base.json 2> $O/base.err && python3 tests/showbench.py $O/base.json
CEDR_BENCH_PERTURB=none timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/noqp.json 2> $O/noqp.err && { echo "== perturb none"; python3 tests/showbench.py $O/noqp.json | head -4; }
CEDR_BENCH_PERTURB=small timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/small.json 2> $O/small.err && { echo "== perturb small"; python3 tests/showbench.py $O/small.json | head -4; }
echo "== stamps"; timeout -k 10 300 python3 tools/stamps.py > $O/stamps.txt 2>&1; cat $O/stamps.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F64"; do
  tag=$(echo $set | md5sum | cut -c1-6)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$R/$O/pmc_$tag" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
done
python3 tools/pmc_table.py $O > $O/pmc_table.txt 2>&1; cat $O/pmc_table.txt
