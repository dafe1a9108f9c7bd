set -u
python tools/gemmbench.py > gpurun_out/r3_gemmbench.json 2> gpurun_out/r3_gemmbench.err
: > gpurun_out/r3_targetdet.jsonl
python tools/bench_targetdet.py --images 3 2>/dev/null | tail -1 >> gpurun_out/r3_targetdet.jsonl
python tools/bench_targetdet.py --images 2 --step-two 2>/dev/null | tail -1 >> gpurun_out/r3_targetdet.jsonl
python tools/bench_targetdet.py --images 3 --no-teacher-stream 2>/dev/null | tail -1 >> gpurun_out/r3_targetdet.jsonl
COIN_TEXT_GRAPH=0 python tools/bench_targetdet.py --images 3 2>/dev/null | tail -1 >> gpurun_out/r3_targetdet.jsonl
python tools/bench_targetdet.py --config bdd100k_rn101 --images 8 --warmup 28 2>/dev/null | tail -1 >> gpurun_out/r3_targetdet.jsonl
python tools/bench_targetdet.py --config swint_fpn --images 3 --warmup 16 2>/dev/null | tail -1 >> gpurun_out/r3_targetdet.jsonl
python tools/bench_targetdet.py --config rn101_fpn --images 4 --warmup 16 2>/dev/null | tail -1 >> gpurun_out/r3_targetdet.jsonl
COIN_FORCE_DDP=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3_bench_1rank_rccl.json
COIN_TEXT_GRAPH=0 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3_bench_text_graph_off.json
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3_bench_text_graph_on.json
bash tools/profile_bench.sh r3td tools/bench_targetdet.py --images 3 --steps 8 > gpurun_out/prof_r3td.log 2>&1
wc -l gpurun_out/r3_targetdet.jsonl; cut -c1-200 gpurun_out/r3_targetdet.jsonl; cut -c1-160 gpurun_out/r3_bench_1rank_rccl.json gpurun_out/r3_bench_text_graph_off.json gpurun_out/r3_bench_text_graph_on.json
