# targetDET step_one: fresh process vs a process in which a PRETrainer ran first (round 5: 52 vs 64 ms); step_two the same
for args in "" "--after-pretrain 8" "" "--after-pretrain 8" "--step-two" "--step-two --after-pretrain 8"; do
  timeout 300 python tools/bench_targetdet.py --images 3 --steps 24 --warmup 8 $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$args |', d['workload'], round(d['ms_per_step'],2), 'median group', round(d['median_group_ms_per_step'],2), d['groups_ms_per_step_in_order'], 'views/s', round(d['student_views_per_s'],1))"
done
