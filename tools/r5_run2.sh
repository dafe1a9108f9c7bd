mkdir -p gpurun_out/r5j
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r5j/pytest_gpu.log 2>&1; tail -15 gpurun_out/r5j/pytest_gpu.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5j/smoke.log 2>&1; tail -2 gpurun_out/r5j/smoke.log | cut -c1-300
