set -u
: > gpurun_out/r3_targetdet.jsonl
python tools/bench_targetdet.py --images 3 --steps 48 2>/dev/null | tail -1 >> gpurun_out/r3_targetdet.jsonl
python tools/bench_targetdet.py --images 3 --steps 48 --no-prefetch 2>/dev/null | tail -1 >> gpurun_out/r3_targetdet.jsonl
python tools/bench_targetdet.py --images 2 --steps 48 --step-two 2>/dev/null | tail -1 >> gpurun_out/r3_targetdet.jsonl
python tools/bench_targetdet.py --images 3 --no-prefetch --no-teacher-stream 2>/dev/null | tail -1 >> gpurun_out/r3_targetdet.jsonl
COIN_TEXT_GRAPH=0 python tools/bench_targetdet.py --images 3 --no-prefetch 2>/dev/null | tail -1 >> gpurun_out/r3_targetdet.jsonl
python tools/bench_targetdet.py --config bdd100k_rn101 --images 8 --warmup 28 2>/dev/null | tail -1 >> gpurun_out/r3_targetdet.jsonl
python tools/bench_targetdet.py --config bdd100k_rn101 --images 8 --warmup 28 --no-prefetch 2>/dev/null | tail -1 >> gpurun_out/r3_targetdet.jsonl
python tools/bench_targetdet.py --config swint_fpn --images 3 --warmup 16 2>/dev/null | tail -1 >> gpurun_out/r3_targetdet.jsonl
python tools/bench_targetdet.py --config rn101_fpn --images 4 --warmup 16 2>/dev/null | tail -1 >> gpurun_out/r3_targetdet.jsonl
bash tools/profile_bench.sh r3td tools/bench_targetdet.py --images 3 --steps 8 > gpurun_out/prof_r3td.log 2>&1
python3 - <<PY
import json
for l in open("gpurun_out/r3_targetdet.jsonl"):
    d=json.loads(l); print(d["config"], d["workload"][10:], d["images_per_step"], round(d["ms_per_step"],1), round(d["median_group_ms_per_step"],1), round(d["fastest_group_ms_per_step"],1), d["groups_ms_per_step_in_order"][:12])
PY
head -1 gpurun_out/prof_r3td/steady_top.txt
