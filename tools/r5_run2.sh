mkdir -p gpurun_out/r5ab
LAB_SHAPES="l2.x.conv1 l2.x.conv2 l2.x.conv3 l4.0.conv1" timeout 300 ./tools/gemm_lab wbench 20 > gpurun_out/r5ab/lab_wbench.log 2>&1; cat gpurun_out/r5ab/lab_wbench.log | cut -c1-330
timeout 1500 python -m pytest tests/test_graphs_gpu.py -q -m gpu > gpurun_out/r5ab/pytest_graphs.log 2>&1; tail -5 gpurun_out/r5ab/pytest_graphs.log
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_graphs_gpu.py > gpurun_out/r5ab/pytest_all.log 2>&1; tail -5 gpurun_out/r5ab/pytest_all.log
