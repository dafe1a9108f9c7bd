for rs in 0 1; do
  COIN_ROLE_STREAMS=$rs COIN_FORCE_DDP=1 COIN_STEP_GRAPHS=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rccl-1rank graphs=1 role_streams=$rs', round(d['ms_per_step'],3), round(d['value'],2), d['config'].get('role_streams'))"
done
COIN_GRAD_ARENA=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('arena only (no process group)', round(d['ms_per_step'],3), round(d['value'],2))"
