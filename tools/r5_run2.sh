mkdir -p gpurun_out/r5g
timeout 900 python -m pytest tests/test_graphs_gpu.py -x -q > gpurun_out/r5g/pytest_graphs.log 2>&1; tail -4 gpurun_out/r5g/pytest_graphs.log
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "storage_rounding" -s > gpurun_out/r5g/pytest_round.log 2>&1; grep -E "res5 bf16|head bf16|passed|failed|^E  " gpurun_out/r5g/pytest_round.log | head -40
td() { name=$1; shift; env "$@" timeout 600 python tools/bench_targetdet.py $ARGS > gpurun_out/r5g/td_$name.log 2>&1; echo "$name rc=$? $(grep -o '"ms_per_step": [0-9.]*\|"groups_ms_per_step_in_order": \[[^]]*\]\|"student_views_per_s": [0-9.]*' gpurun_out/r5g/td_$name.log | tr '\n' ' ')"; }
ARGS="--images 3"
td one_default X=1
ARGS="--images 3 --no-teacher-stream"
td one_nostream X=1
ARGS="--images 3 --no-prefetch"
td one_noprefetch X=1
ARGS="--images 3"
td one_tg_always COIN_TEACHER_GRAPH=always
timeout 900 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r5g/bench.json 2> gpurun_out/r5g/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5g/bench.json").read().strip().splitlines()[-1])
print("bench", round(d["value"],2), round(d["ms_per_step"],2), d["config"].get("host_enqueue_ms"), d["config"].get("step_graphs"), "roof", round(d["roofline"]["frac"],4), "sec", d.get("secondary",{}).get("value"), d.get("secondary",{}).get("groups_ms_per_step_in_order"))
for r in d["kernels"]["coin_conv_gemm_bf16"].get("shapes", [])[:30]: print(r)
PY
