mkdir -p gpurun_out/r5td
for g in 1 0; do COIN_STEP_GRAPHS=$g timeout 600 python tools/host_timeline.py --images 3 > gpurun_out/r5td/timeline_g$g.log 2>&1; echo "== graphs=$g"; grep -v Warning gpurun_out/r5td/timeline_g$g.log | tail -25 | cut -c1-200; done
for g in 1 0; do COIN_STEP_GRAPHS=$g timeout 600 python tools/bench_targetdet.py --images 3 > gpurun_out/r5td/td_g$g.log 2>&1; tail -1 gpurun_out/r5td/td_g$g.log | cut -c1-330; done
