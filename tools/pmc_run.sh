#!/bin/bash
# HBM-traffic counters for the hand-written streaming kernels (run on the GPU box):  bash tools/pmc_run.sh <tag>
# Two separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only, as the guide prescribes.
set -u
tag=${1:-r1}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_target.py > /tmp/pmc_$ctr.log 2>&1
  f=$(find /tmp/pmc_$ctr -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$f" "$out/pmc_$ctr.csv"; else echo "no counter csv for $ctr"; tail -5 /tmp/pmc_$ctr.log; fi
done
