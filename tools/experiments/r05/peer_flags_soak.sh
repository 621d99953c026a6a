#!/bin/bash
# Soak of the peer-flag transport: tests/peer_flags_worker.py with every loop PEER_REPEAT times longer, ranks as
# processes on GPU 0; every collective's result is compared bitwise with the oracle's. Output: gpurun_out/r05pf/soak.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
O=gpurun_out/r05pf
mkdir -p $O
{
echo "# kernels_sha256 $(python3 -c 'import compose_amd as c; print(c.kernels_sha256())')"
IFS=';' read -ra CASES <<< "${PEER_SOAK_CASES:-4 100 4096 1531;2 100 64 700;3 100 4096 20011;4 200 128 333}"
for cfg in "${CASES[@]}"; do
  set -- $cfg
  D=$(mktemp -d /tmp/pfsXXXXXX)
  pids=""
  for r in $(seq 0 $(($1-1))); do
    PEER_REPEAT=$2 PEER_SLOT_DOUBLES=$3 PEER_NCELLS=$4 timeout -k 10 400 python3 tests/peer_flags_worker.py $r $1 $D > $O/soak_$r.txt 2> $O/soak_$r.err &
    pids="$pids $!"
  done
  rc=0
  for p in $pids; do wait $p || rc=1; done
  for r in $(seq 0 $(($1-1))); do tail -1 $O/soak_$r.txt; done
  [ $rc = 0 ] || { echo "FAILED: $cfg"; tail -5 $O/soak_0.err; exit 1; }
done
} | tee $O/soak.txt
