python -m pytest tests/test_gpu_qlt.py tests/test_gpu_threaded_ranks.py tests/test_gpu_large.py -x -q -m gpu 2>&1 | tail -1
for n in 1500000 187500; do for i in 1 2; do
python bench.py --ncells $n --serial --no-cpu-baseline --steps 40 > gpurun_out/sd.json 2>/dev/null && python - $n <<'PY'
import json,sys
d=json.load(open("gpurun_out/sd.json"))
print(sys.argv[1], "qlt %.4f ms"%d["qlt"]["ms_per_run"], "step %.4f"%d["ms_per_step"], " ".join("%s/%d %.4f"%(k["kernel"][:8],k["launch"],k["avg_ms"]) for k in d["kernels"] if k["kernel"].startswith(("qlt_down","qlt_top","tree_up")) and k["launch"]<8192))
PY
done; done
python tools/chain_rank0_of_8.py 2>&1 | grep -v amdgpu | head -3
