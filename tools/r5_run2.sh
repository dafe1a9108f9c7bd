mkdir -p gpurun_out/r5ah
timeout 600 ./tools/gemm_lab check > gpurun_out/r5ah/lab_check.log 2>&1; tail -1 gpurun_out/r5ah/lab_check.log
LAB_SHAPES="l3.x.conv1 l3.x.conv3 l3.x.conv2" timeout 300 ./tools/gemm_lab bench 30 > gpurun_out/r5ah/lab_bench_l3.log 2>&1; cut -c1-700 gpurun_out/r5ah/lab_bench_l3.log
