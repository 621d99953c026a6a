#!/bin/bash
# Round 5 experiment: ONE launch for the three problem classes of a mixed list (merged_class_launch.patch, built out of tree
# into compose_amd/libcedr_hip_merged.so) against the committed launch per class. Parity first, then bench.py --types mix.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
mkdir -p gpurun_out/r05m
M=$R/compose_amd/libcedr_hip_merged.so
CEDR_HIP_LIB=$M CEDR_HIP_DOWN_CLS_MIN_TILES=1 timeout -k 10 500 python3 -m pytest tests/test_gpu_qlt.py tests/test_gpu_large.py::test_mixed_types_class_forms_at_scale -m gpu -x -q > gpurun_out/r05m/parity.txt 2>&1 || { tail -20 gpurun_out/r05m/parity.txt; exit 1; }
tail -2 gpurun_out/r05m/parity.txt
for rep in 1 2 3; do
for cfg in "A=committed" "CEDR_HIP_LIB=$M" "CEDR_HIP_LIB=$M CEDR_HIP_DOWN_CLS=2"; do
  env $cfg timeout -k 10 200 python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --types mix > gpurun_out/r05m/m.json 2> gpurun_out/r05m/m.err || { tail -3 gpurun_out/r05m/m.err; exit 1; }
  python3 - "$cfg" <<'PY' | tee -a gpurun_out/r05m/ab.txt
import json,sys
d=json.load(open("gpurun_out/r05m/m.json"))
ks=" ".join("%s/%d=%.4f" % (k["kernel"].replace("_kernel",""), k["launch"], k["avg_ms"]) for k in d["kernels"] if k["kernel"] in ("qlt_down_kernel",))
print("%-40s qlt %.4f | %s" % (sys.argv[1].replace("/root/repo/compose_amd/","").split("compose_amd/")[-1], d["qlt"]["ms_per_run"], ks))
PY
done
done
