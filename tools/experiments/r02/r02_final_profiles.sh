#!/bin/bash
# Round-2 evidence run on the GPU box: bench line, rocprofv3 kernel trace + stats, HBM traffic (PMC,
# separate passes), SQ/VALU/LDS counters. Copy the summaries it prints under profiles/ (r02_*).
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r02final
rm -rf $O; mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python3 tests/showbench.py $O/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/kt" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_kt.json 2> $O/bench_kt.err; echo "kernel trace rc=$?"
python3 tools/kernel_stats_by_grid.py $O/kt/*/*kernel_trace.csv $O/kernel_stats_by_grid.csv | head -14
cp $O/kt/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$R/$O/prof_traffic/$c" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/traffic_$c.log 2>&1 || echo "pmc $c failed"
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$R/$O/prof_traffic/odd_$c" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --ncells 1500001 > $O/traffic_odd_$c.log 2>&1 || echo "pmc odd $c failed"
done
python3 tools/traffic_summary.py $O/prof_traffic | tee $O/traffic_summary.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F64"; do
  tag=$(echo $set | md5sum | cut -c1-6)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$R/$O/pmc_$tag" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
done
python3 tools/pmc_table.py $O $O/valu.json > $O/pmc_table.txt 2>&1; grep -A2 "qlt_down_sp_kernel grid=2097152\|tree_up_micro_kernel grid=2097152\|caas_reduce_kernel" $O/pmc_table.txt | cut -c1-400
rm -rf $O/kt $O/prof_traffic/*/*/*agent_info.csv
