#!/bin/bash
# Run a python command under rocprofv3 --kernel-trace --stats on the GPU box and leave the CSVs under
# gpurun_out/<tag>/prof.   usage: scripts/gpu_prof.sh <tag> <python args...>
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o trace -- python "$@" > "$OUT/prof.log" 2>&1
echo "rc=$?"
tail -3 "$OUT/prof.log"
find "$OUT/prof" -name '*.csv' | head
