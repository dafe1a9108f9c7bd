# 1-rank RCCL (COIN_FORCE_DDP=1): the default bench with the step graphs on / off, interleaved; and without a process group for reference
mkdir -p gpurun_out/r6j
for r in 1 2; do
  for g in 1 0; do
    COIN_FORCE_DDP=1 COIN_STEP_GRAPHS=$g timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rccl-1rank graphs=$g', round(d['ms_per_step'],3), round(d['value'],2), d['config']['gpu_telemetry']['sclk_mhz'])"
  done
done
timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no process group graphs=1', round(d['ms_per_step'],3), round(d['value'],2))"
