#!/bin/bash
# BASELINE.json's other configurations on one GPU (the bench line's workload is config C): B = CAAS 150k cells x 10 tracers;
# the 8-way shard of D = 750k cells x 128 tracers; D's tracer count at C's cell count.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/other
echo "# HEAD $(cat .head_sha 2>/dev/null | tr '\n' ' ') kernels_sha256 $(python3 -c 'import compose_amd as c; print(c.kernels_sha256())' 2>/dev/null)"
for cfg in "150000 10" "750000 128" "1500000 128"; do
  set -- $cfg
  timeout -k 10 400 python3 bench.py --ncells $1 --ntracers $2 --serial --no-cpu-baseline --check --steps 40 > gpurun_out/other/b.json 2> gpurun_out/other/b.err || { tail -3 gpurun_out/other/b.err; continue; }
  python3 - $1 $2 <<'PY'
import json,sys
d=json.load(open("gpurun_out/other/b.json")); n,nt=int(sys.argv[1]),int(sys.argv[2])
q,c=d["qlt"]["ms_per_run"],d["caas"]["ms_per_run"]
print("%d cells x %d tracers (one stream): QLT::run %.4f ms = %.2f TB/s on 64.4 B per cell-tracer (%.1f%% of 8 TB/s); CAAS::run %.4f ms = %.2f TB/s on 64 B (%.1f%%); step %.4f ms; check %s" %
      (n, nt, q, n*nt*(64.0+16.0/nt)/q/1e9, 100*n*nt*(64.0+16.0/nt)/q/1e9/8, c, n*nt*64.0/c/1e9, 100*n*nt*64.0/c/1e9/8, d["ms_per_step"], d.get("check_ok")))
PY
done
# the dependent-chain floor of the small configurations next to their measured times
timeout -k 10 300 python3 tools/chain_floor.py 2>&1 | grep -v amdgpu.ids
