mkdir -p gpurun_out/r5dbg4
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_graphs_gpu.py -x -q -s -m gpu > gpurun_out/r5dbg4/e2e_graphs_s.log 2>&1; echo "e2e+graphs rc=$?"; grep -n -i -m3 "fault\|abort" gpurun_out/r5dbg4/e2e_graphs_s.log | cut -c1-200; tail -2 gpurun_out/r5dbg4/e2e_graphs_s.log | cut -c1-200
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r5dbg4/pytest_gpu.log 2>&1; echo "full suite rc=$?"; grep -i -m3 "fault\|abort" gpurun_out/r5dbg4/pytest_gpu.log; tail -3 gpurun_out/r5dbg4/pytest_gpu.log | cut -c1-200
