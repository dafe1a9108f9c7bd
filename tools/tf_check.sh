export COIN_TEACHER_FIRST=1
COIN_FORCE_DDP=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TF rccl-1rank', round(d['ms_per_step'],3))"
timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TF no pg', round(d['ms_per_step'],3))"
for args in "--after-pretrain 8" "" "--after-pretrain 8" ""; do
  timeout 300 python tools/bench_targetdet.py --images 3 --steps 24 --warmup 8 $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TF $args |', d['workload'], round(d['ms_per_step'],2), d['groups_ms_per_step_in_order'])"
done
