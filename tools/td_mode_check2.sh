for rs in 0; do
for args in "" "--after-pretrain 8" "--step-two" ""; do
  COIN_ROLE_STREAMS=$rs timeout 300 python tools/bench_targetdet.py --images 3 --steps 24 --warmup 8 $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('role_streams=$rs $args |', d['workload'], round(d['ms_per_step'],2), 'median group', round(d['median_group_ms_per_step'],2), d['groups_ms_per_step_in_order'])"
done
done
