#!/bin/bash
# Builds and runs peer_flag_probe.cpp: an owner and a writer process on the one GPU. Output: gpurun_out/r05pf/probe.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out/r05pf
D=$(mktemp -d /tmp/pfXXXXXX)
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -o $D/probe tools/experiments/r05/peer_flag_probe.cpp 2> gpurun_out/r05pf/build.err || { tail -5 gpurun_out/r05pf/build.err; exit 1; }
timeout -k 5 60 $D/probe owner $D > gpurun_out/r05pf/owner.txt 2>&1 &
OWNER=$!
timeout -k 5 45 $D/probe writer $D > gpurun_out/r05pf/writer.txt 2>&1
echo "writer rc $?" >> gpurun_out/r05pf/writer.txt
wait $OWNER
echo "owner rc $?" >> gpurun_out/r05pf/owner.txt
cat gpurun_out/r05pf/owner.txt gpurun_out/r05pf/writer.txt | tee gpurun_out/r05pf/probe.txt
