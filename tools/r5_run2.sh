mkdir -p gpurun_out/r5ag
timeout 600 ./tools/gemm_lab wcheck > gpurun_out/r5ag/lab_wcheck.log 2>&1; grep -c OK gpurun_out/r5ag/lab_wcheck.log; tail -1 gpurun_out/r5ag/lab_wcheck.log
LAB_SHAPES="l2.x.conv1 l2.x.conv2 l2.x.conv3 l4.0.conv1 l3.x.conv2 rpn.conv" timeout 300 ./tools/gemm_lab wbench 20 > gpurun_out/r5ag/lab_wbench.log 2>&1; sed 's/, "sliced_ms".*/}/' gpurun_out/r5ag/lab_wbench.log | cut -c1-330
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wgrad or fan_in" > gpurun_out/r5ag/pytest_wgrad.log 2>&1; tail -2 gpurun_out/r5ag/pytest_wgrad.log
timeout 1500 python -m pytest tests/test_graphs_gpu.py -q -m gpu > gpurun_out/r5ag/pytest_graphs.log 2>&1; tail -4 gpurun_out/r5ag/pytest_graphs.log
timeout 600 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r5ag/bench.log 2>&1; tail -1 gpurun_out/r5ag/bench.log | cut -c1-250
