#!/bin/bash
# gpurun wrapper: records the HEAD the snapshot was taken at in .head_sha (git-ignored, shipped), so that
# tools running on the GPU box (no .git there) can stamp the files they write. Usage: tools/gpu.sh <timeout> '<command>'
cd "$(dirname "$0")/.."
{ git rev-parse HEAD; git status --porcelain | grep -q . && echo "+uncommitted changes"; } > .head_sha 2>/dev/null
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
