#!/bin/bash
# Round 5: what the peer-flag transport costs a step on ONE GPU, next to the single-rank communicator and real librccl --
# bench.py --pretend-ranks 8 (rank 0 of 8, 187.5k-cell shard x 40 tracers) over a communicator of ONE rank of each kind,
# alternating, two passes; then real rank PROCESSES sharing GPU 0 (the kernels of the ranks contend for the one GPU: these
# rows show that the transport runs at bench sizes, not a speed). Output: gpurun_out/r05pf/cost.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r05pf
mkdir -p $O
show() { python3 - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("step %.4f ms  QLT::run %.4f ms  CAAS::run %.4f ms  check %s | %s" % (d["ms_per_step"], d["qlt"]["ms_per_run"], d["caas"]["ms_per_run"], d.get("check_ok"), d["config"]["transport"][:60]))
PY
}
{
echo "# kernels_sha256 $(python3 -c 'import compose_amd as c; print(c.kernels_sha256())')"
for pass in 1 2; do
  for st in "" "--serial"; do
    for rc in "" "--force-rccl" "--comm peer-flags"; do
      timeout -k 10 200 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --check --pretend-ranks 8 $rc $st > $O/c.json 2> $O/c.err && {
        echo -n "rank 0 of 8 (rehearsal), ${rc:-self communicator}, ${st:-two streams}: "; show $O/c.json; } || { tail -3 $O/c.err; exit 1; }
    done
  done
done
for n in 2 4; do
  for st in "" "--serial"; do
    timeout -k 10 300 python3 bench.py --gpus $n --same-gpu --comm peer-flags --steps 50 --warmup 5 --no-cpu-baseline --check --ncells $((187500*n)) $st > $O/c.json 2> $O/c.err && {
      echo -n "$n rank processes on GPU 0, 187.5k cells each, peer-flags, ${st:-two streams}: "; show $O/c.json; } || { tail -5 $O/c.err; exit 1; }
  done
done
} | tee $O/cost.txt
