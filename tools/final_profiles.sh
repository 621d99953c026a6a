#!/bin/bash
# Evidence run of round ${RN} on the GPU box, ONE script for every file under profiles/${RN}_* so that all of them
# come from the same HEAD (recorded in each): bench line, rocprofv3 kernel trace + stats, HBM traffic (PMC,
# separate passes), SQ counters, shard sizes and the rank-0-of-8 chain (incl. level mode), CAAS + BFB reducer,
# top-kernel phase stamps, host setup times. Start it through tools/gpu.sh (which writes .head_sha); copy
# gpurun_out/${RN}final/* to profiles/ with the round's prefix afterwards (tools/collect_profiles.sh).
RN=${RN:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/${RN}final
# PART=A: the bench line, kernel trace, counters; PART=B: shards, pairings, stamps, setup, mixed types, other configurations
# (one gpurun call each: together they take longer than a call may); default: both
PART=${PART:-AB}
case $PART in *A*) rm -rf $O;; esac
mkdir -p $O
HEAD=$(cat .head_sha 2>/dev/null | tr '\n' ' ')
# sha256 of the cedr_kernels.hip the library under test was built from (stamped in at build time): every file written
# below carries it next to the HEAD, and bench.py refuses a traffic file whose sha is not the running library's
export KSHA=$(python3 -c 'import compose_amd as c; print(c.kernels_sha256())' 2>/dev/null)
HEAD="$HEAD kernels_sha256 $KSHA"
echo "HEAD $HEAD" | tee $O/HEAD.txt
case $PART in *A*)
timeout -k 10 400 python3 bench.py --check > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python3 tests/showbench.py $O/bench.json
# (the profiling passes below run with --serial: one stream, as bench.py's own roofline leg does, so that a kernel's
# duration and counters are not stretched by a concurrent kernel of the other CDR)
timeout -k 10 400 python3 bench.py --serial --check --no-cpu-baseline > $O/bench_serial.json 2> $O/bench_serial.err; echo "serial: $(python3 tests/showbench.py $O/bench_serial.json | head -1)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/kt" -- python3 bench.py --serial --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_kt.json 2> $O/bench_kt.err; echo "kernel trace rc=$?"
python3 tools/kernel_stats_by_grid.py $O/kt/*/*kernel_trace.csv $O/kernel_stats_by_grid.csv | head -16
cp $O/kt/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$R/$O/prof_traffic/$c" -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline > $O/traffic_$c.log 2>&1 || echo "pmc $c failed"
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$R/$O/prof_traffic/odd_$c" -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --ncells 1500001 > $O/traffic_odd_$c.log 2>&1 || echo "pmc odd $c failed"
done
python3 tools/traffic_summary.py $O/prof_traffic | tee $O/traffic_summary.txt
# ... and of the production pairing: CAAS fused with the BFB tree reducer's first stage (bench.py --caas-bfb 4)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$R/$O/prof_traffic_bfb/$c" -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --caas-bfb 4 > $O/traffic_bfb_$c.log 2>&1 || echo "pmc bfb $c failed"
done
python3 tools/traffic_summary.py $O/prof_traffic_bfb | tee $O/traffic_bfb_summary.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F64"; do
  tag=$(echo $set | md5sum | cut -c1-6)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$R/$O/pmc_$tag" -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
done
python3 tools/pmc_table.py $O $O/valu.json > $O/pmc_table.txt 2>&1; grep -A2 "qlt_down_sp_kernel grid=2097152\|tree_up_micro_kernel grid=2097152\|caas_reduce_kernel" $O/pmc_table.txt | cut -c1-400
;; esac
case $PART in *B*)
{ echo "# HEAD $HEAD"; RN=$RN bash tools/shards.sh 2>&1 | grep -v amdgpu.ids; } | tee $O/shards_and_chain.txt
RN=$RN timeout -k 10 400 python3 tools/caas_bfb_profile.py 2>&1 | grep -v amdgpu.ids | tee $O/caas_bfb.txt
cp gpurun_out/${RN}_caas_bfb.json $O/caas_bfb.json; cp gpurun_out/${RN}_caas_bfb.csv $O/caas_bfb.csv
if [ -f compose_amd/variants/topstamps.so ]; then { echo "# HEAD $HEAD"; timeout -k 10 200 python3 tools/top_stamps.py 2>&1 | grep -v amdgpu.ids; } | tee $O/top_stamps.txt; fi
{ echo "# HEAD $HEAD; host: $(lscpu | grep 'Model name' | sed 's/.*: *//')"; timeout -k 10 300 python3 tools/setup_time.py 1500000 6000000 2>&1 | grep -v amdgpu.ids; } | tee $O/setup_time.txt
# ... and of the mixed tracer list (bench.py --types mix): the root->leaves kernel for any tracer types
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$R/$O/prof_traffic_mix/$c" -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --types mix > $O/traffic_mix_$c.log 2>&1 || echo "pmc mix $c failed"
done
python3 tools/traffic_summary.py $O/prof_traffic_mix | tee $O/traffic_mix_summary.txt
{ echo "# HEAD $HEAD: bench.py --types mix (cst, st, ct, t, nn, cnn cycling over 40 tracers)"; bash tools/experiments/r03/mix.sh "CEDR_X=0" 2>&1 | grep -v amdgpu.ids;
  echo "# HBM traffic of the same run's kernels (PMC, separate passes):"; cat $O/traffic_mix_summary.txt; } | tee $O/mix.txt
{ RN=$RN bash tools/other_configs.sh 2>&1 | grep -v amdgpu.ids; } | tee $O/other_configs.txt
;; esac
rm -rf $O/kt $O/prof_traffic*/*/*/*agent_info.csv $O/pmc_*/*/*agent_info.csv
du -sh $O
