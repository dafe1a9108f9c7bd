for rs in 0 1; do
  COIN_REDUCER_RESLICE=$rs COIN_FORCE_DDP=1 COIN_STEP_GRAPHS=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rccl-1rank graphs=1 reslice=$rs', round(d['ms_per_step'],3), round(d['value'],2))"
  COIN_REDUCER_RESLICE=$rs COIN_FORCE_DDP=1 COIN_STEP_GRAPHS=0 timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rccl-1rank graphs=0 reslice=$rs', round(d['ms_per_step'],3), round(d['value'],2))"
done
