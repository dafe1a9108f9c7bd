mkdir -p gpurun_out/r6h
for P in 0 1 2 3 4 5 6 7; do
  PRE_STREAMS=$P timeout 200 python tools/bench_targetdet.py --images 3 --steps 16 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PRE_STREAMS=$P', round(d['ms_per_step'],2), d['groups_ms_per_step_in_order'])"
done
for P in 0 3 4; do
  GPU_MAX_HW_QUEUES=8 PRE_STREAMS=$P timeout 200 python tools/bench_targetdet.py --images 3 --steps 16 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('Q8 PRE_STREAMS=$P', round(d['ms_per_step'],2), d['groups_ms_per_step_in_order'])"
done
