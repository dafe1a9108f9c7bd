#!/bin/bash
# HBM-traffic counters of the bench's own launch mix (run on the GPU box):  bash tools/pmc_bench.sh <tag>
# Two passes over a short bench run (3 steps, no warm-up) with the step graphs OFF (COIN_STEP_GRAPHS=0, exported below: bench.py runs 5 more
# steps after its timed region, enough for both stretches to be captured and replayed -- the per-kernel means would mix eager, dry-run and
# replayed launches; round-5 ADVICE): every launch counted here is an eager one of the same kernels.
# Per-kernel totals go to gpurun_out/prof_<tag>/pmc_bench_<COUNTER>.csv
set -u
tag=${1:-r1}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export COIN_STEP_GRAPHS=0
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$ctr
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmcb_$ctr -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-secondary > /tmp/pmcb_$ctr.log 2>&1
  f=$(find /tmp/pmcb_$ctr -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$f" "$out/pmc_bench_$ctr.csv" | grep -E "bn_bwd|roi_align|conv_gemm" ; else echo "no counter csv for $ctr"; tail -5 /tmp/pmcb_$ctr.log; fi
done
