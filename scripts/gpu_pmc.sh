#!/bin/bash
# Collect one PMC counter set per pass (rocprofv3 --pmc, no trace domains) for a python command.
# usage: scripts/gpu_pmc.sh <tag> "<counters pass1>" "<counters pass2>" -- <python args...>
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
PASSES=()
while [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for P in "${PASSES[@]}"; do
  PYTHONPATH=$R rocprofv3 --pmc $P --output-format csv -d "$OUT/pmc$i" -o pmc -- python "$@" > "$OUT/pmc$i.log" 2>&1
  echo "pass $i ($P) rc=$?"
  i=$((i+1))
done
find "$OUT" -name '*.csv' | head -20
