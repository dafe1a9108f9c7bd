mkdir -p gpurun_out/r5ddp
timeout 900 python -m pytest tests/test_ddp_gpu.py tests/test_graphs_gpu.py -q -m gpu > gpurun_out/r5ddp/pytest.log 2>&1; tail -3 gpurun_out/r5ddp/pytest.log | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r5ddp/bench_plain.log 2>&1; tail -1 gpurun_out/r5ddp/bench_plain.log | cut -c1-230
COIN_FORCE_DDP=1 timeout 600 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r5ddp/bench_ddp1.log 2>&1; tail -1 gpurun_out/r5ddp/bench_ddp1.log | cut -c1-230
COIN_FORCE_DDP=1 COIN_STEP_GRAPHS=0 timeout 600 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r5ddp/bench_ddp1_eager.log 2>&1; tail -1 gpurun_out/r5ddp/bench_ddp1_eager.log | cut -c1-230
