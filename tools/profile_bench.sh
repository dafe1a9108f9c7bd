#!/bin/bash
# Steady-state kernel profile of the default bench (run on the GPU box):  bash tools/profile_bench.sh <tag>
#   or of another script of this repo:                                    bash tools/profile_bench.sh <tag> tools/bench_targetdet.py --images 3
# Writes small summaries to gpurun_out/prof_<tag>/ (copy the ones to keep into profiles/).
set -u
tag=${1:-r1}
shift || true
if [ $# -gt 0 ]; then prog=$1; shift; args="$*"; else prog=bench.py; args="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary"; fi
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o b -- python3 $GRAFT_REPO_ROOT/$prog $args > "$out/bench_under_rocprof.log" 2>&1
tail -1 "$out/bench_under_rocprof.log" | cut -c1-400
stats=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
trace=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
if [ -n "$stats" ]; then
  head -1 "$stats" > "$out/bench_kernel_stats_ours.csv"
  grep -E "roi_align|bn_|gemm_nt|conv_gemm|conv_wgrad|wgrad_reduce|tn_reduce|slab_sum|conv_stats|mil_focal|nms_|sgd_kernel|ema_kernel|cosine_|mil_ce|kl_div|box_reg|l1_mean|rpn_losses|normalize_pad|avgpool2|transpose_kernel|bias_act|bias_sum|weight_dgrad|window_attn|anchor_match|sample_labels" "$stats" >> "$out/bench_kernel_stats_ours.csv"
  head -61 "$stats" > "$out/bench_kernel_stats_top60.csv"
fi
if [ -n "$trace" ]; then
  python3 $GRAFT_REPO_ROOT/tools/trace_summary.py "$trace" "$out/steady_kernel_summary.csv" --window-ms 300 --sgd-per-step $([[ "$prog" == *targetdet* ]] && echo 2 || echo 1) --list p8_kernel > "$out/steady_top.txt" 2>&1
  head -120 "$out/steady_top.txt"
fi
