mkdir -p gpurun_out/r5m
timeout 900 python -m pytest tests/test_graphs_gpu.py -q > gpurun_out/r5m/pytest_graphs.log 2>&1; tail -3 gpurun_out/r5m/pytest_graphs.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/r5m/bench_line.json 2> gpurun_out/r5m/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5m/bench_line.json").read().strip().splitlines()[-1])
print("bench", round(d["value"],2), round(d["ms_per_step"],2), "loss", d["config"]["final_loss"], d["config"].get("step_graphs"), "roof", round(d["roofline"]["frac"],4), "sec", d.get("secondary",{}).get("value"), d.get("secondary",{}).get("groups_ms_per_step_in_order"), d.get("secondary",{}).get("final_loss"), "cpu", d.get("cpu_baseline",{}).get("value"))
PY
bash tools/profile_bench.sh r5 > gpurun_out/r5m/profile.log 2>&1; head -12 gpurun_out/prof_r5/steady_top.txt | cut -c1-160
COIN_STEP_GRAPHS=0 bash tools/pmc_bench.sh r5 > gpurun_out/r5m/pmc.log 2>&1; tail -12 gpurun_out/r5m/pmc.log | cut -c1-200
