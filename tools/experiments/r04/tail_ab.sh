#!/bin/bash
# A/B of the first-stage launches' tail split (CEDR_HIP_TAIL = workgroups per tail tile; 0 = off) on one box.
out=gpurun_out/r04_tail; mkdir -p $out
CEDR_HIP_TAIL=2 timeout -k 10 600 python -m pytest tests/test_gpu_qlt.py tests/test_gpu_large.py tests/test_gpu_bfb.py -x -q -m gpu > $out/pytest.log 2>&1 || { tail -20 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
CEDR_HIP_TAIL_VERBOSE=1 CEDR_HIP_TAIL=2 python bench.py --serial --no-cpu-baseline --steps 5 2>&1 >/dev/null | grep "tail split" | sort | uniq -c
for i in 1 2; do
  for v in "0 0" "2 0" "2 1" "5 0" "3 1"; do
    set -- $v
    CEDR_HIP_TAIL=$1 CEDR_HIP_TAIL_ROUNDS=$2 timeout -k 10 300 python bench.py --serial --no-cpu-baseline --steps 40 > $out/bench_$1_$2_$i.json 2>> $out/err.log || exit 1
    python tests/showbench.py $out/bench_$1_$2_$i.json | head -4 | sed "s/^/tail k=$1 r=$2: /"
  done
done
