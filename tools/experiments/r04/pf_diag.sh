#!/bin/bash
# One diagnostic pass: the register-prefetch build at 1.5M cells, first without the placement tries (real data only),
# then with them (the timed sweeps run on zeros). stderr kept.
out=gpurun_out/r04_pfdiag; mkdir -p $out
CEDR_HIP_PLACE_TRIES=1 CEDR_HIP_DOWN_PF=100000 timeout -k 10 120 python bench.py --serial --no-cpu-baseline --steps 5 --check > $out/a.json 2> $out/a.err; echo "tries=1 rc=$?"; tail -3 $out/a.err; python tests/showbench.py $out/a.json | head -3 &&
CEDR_HIP_PLACE_TRIES=4 CEDR_HIP_DOWN_PF=100000 timeout -k 10 120 python bench.py --serial --no-cpu-baseline --steps 5 --check > $out/b.json 2> $out/b.err; echo "tries=4 rc=$?"; tail -3 $out/b.err
