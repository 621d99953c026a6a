#!/bin/bash
# The randomised soaks in one call (bit-for-bit against the oracle). Writes gpurun_out/${RN:-r05}soak/summary.txt.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
O=gpurun_out/${RN:-r05}soak
mkdir -p $O
echo "# Soaks on one MI355X at HEAD $(cat .head_sha 2>/dev/null | tr '\n' ' ') (bit-for-bit against the oracle)" > $O/summary.txt
run () {  # name, count
  timeout -k 10 1000 python3 tools/$1 $2 > $O/$1.log 2>&1
  local rc=$?
  printf "%-32s %s%s\n" "tools/$1 $2:" "$(tail -1 $O/$1.log)" "$([ $rc -ne 0 ] && echo "  [exit $rc]")" >> $O/summary.txt
  tail -1 $O/$1.log
  return $rc
}
run soak_parity.py 60 && run soak_ranks_trees.py 400 && run soak_ranks.py 80 && run soak_zero_copy.py 200 && run soak_pairing.py 150
cat $O/summary.txt
