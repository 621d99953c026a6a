#!/bin/bash
# Round 5: the two forms of a peer-flag collective -- three stream operations (put launch, hipStreamWaitValue64, take launch)
# against ONE launch that waits in the kernel (CEDR_HIP_PEER_WAIT=kernel): latency back to back (tools/collective_latency.py),
# cost in the rank-0-of-8 rehearsal step, and the soak in the one-launch form. Output: gpurun_out/r05pf/forms.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r05pf
mkdir -p $O
show() { python3 - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("step %.4f ms  QLT::run %.4f ms  CAAS::run %.4f ms  check %s" % (d["ms_per_step"], d["qlt"]["ms_per_run"], d["caas"]["ms_per_run"], d.get("check_ok")))
PY
}
{
echo "# kernels_sha256 $(python3 -c 'import compose_amd as c; print(c.kernels_sha256())')"
echo "## latency, calls back to back on one stream"
for w in stream kernel; do
  echo "# CEDR_HIP_PEER_WAIT=$w"
  for n in 1 2 4; do
    CEDR_HIP_PEER_WAIT=$w timeout -k 10 200 python3 tools/collective_latency.py --ranks $n 2>&1 | grep -E "^(single|rccl,|peer)" || exit 1
  done
done
echo "## rank 0 of 8 (rehearsal, 187.5k cells x 40 tracers), communicator of ONE rank, alternating"
for pass in 1 2; do
  for st in "" "--serial"; do
    for cfg in "self" "stream" "kernel"; do
      if [ $cfg = self ]; then extra=""; else extra="--comm peer-flags"; fi
      CEDR_HIP_PEER_WAIT=$cfg timeout -k 10 200 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --check --pretend-ranks 8 $extra $st > $O/f.json 2> $O/f.err && {
        echo -n "$cfg, ${st:-two streams}: "; show $O/f.json; } || { tail -3 $O/f.err; exit 1; }
    done
  done
done
echo "## real rank processes on GPU 0, 187.5k cells each (the ranks' kernels share the GPU)"
for w in stream kernel; do
  for n in 2 4; do
    CEDR_HIP_PEER_WAIT=$w timeout -k 10 300 python3 bench.py --gpus $n --same-gpu --comm peer-flags --steps 50 --warmup 5 --no-cpu-baseline --check --ncells $((187500*n)) --serial > $O/f.json 2> $O/f.err && {
      echo -n "$n ranks, $w, --serial: "; show $O/f.json; } || { tail -5 $O/f.err; exit 1; }
  done
done
echo "## soak, CEDR_HIP_PEER_WAIT=kernel"
CEDR_HIP_PEER_WAIT=kernel bash tools/experiments/r05/peer_flags_soak.sh | grep -v "^#"
} 2>&1 | tee $O/forms.txt
