#!/bin/bash
# bench.py at N > 1 (two ranks sharing this box's one GPU over the host-staged rehearsal transport): the normal two-stream
# run, and a two-stream run that stalls (test hook): the watchdog must end it with the one-stream line measured before.
python bench.py --steps 5 --no-cpu-baseline > gpurun_out/wd1.json 2> gpurun_out/wd1.err; echo "N=1 rc=$?"; python tests/showbench.py gpurun_out/wd1.json | head -1
python bench.py --gpus 2 --comm gloo --same-gpu --steps 5 --ncells 200000 --no-cpu-baseline > gpurun_out/wd2.json 2> gpurun_out/wd2.err; echo "N=2 rc=$?"
python -c "import json; d=json.load(open('gpurun_out/wd2.json')); print('N=2', d['n_gpus'], d['config']['streams'], round(d['ms_per_step'],3), round(d['serial']['ms_per_step'],3), d.get('note'))"
CEDR_BENCH_SIMULATE_STALL=1 CEDR_BENCH_WATCHDOG_S=8 timeout -k 5 120 python bench.py --gpus 2 --comm gloo --same-gpu --steps 5 --ncells 200000 --no-cpu-baseline > gpurun_out/wd3.json 2> gpurun_out/wd3.err; echo "stalled N=2 rc=$?"
python -c "import json; d=json.load(open('gpurun_out/wd3.json')); print('stalled N=2', d['n_gpus'], d['config']['streams'], round(d['ms_per_step'],3), d.get('note'))"
python -m pytest tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -2
