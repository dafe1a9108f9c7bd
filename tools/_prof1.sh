set -u
python bench.py > gpurun_out/r3_bench_run.log 2>&1; tail -1 gpurun_out/r3_bench_run.log > gpurun_out/r3_bench_line.json
bash tools/profile_bench.sh r3 > gpurun_out/prof_r3.log 2>&1
bash tools/pmc_bench.sh r3 > gpurun_out/pmc_r3.log 2>&1
(timeout 300 tools/gemm_lab check; timeout 300 tools/gemm_lab wcheck) > gpurun_out/r3_lab_check.log 2>&1
timeout 300 tools/gemm_lab bench 20 > gpurun_out/r3_lab_nt_hot.log 2>&1
timeout 300 tools/gemm_lab bench 7 cold > gpurun_out/r3_lab_nt_cold.log 2>&1
timeout 300 tools/gemm_lab wbench 20 > gpurun_out/r3_lab_wgrad_hot.log 2>&1
timeout 300 tools/gemm_lab wbench 7 cold > gpurun_out/r3_lab_wgrad_cold.log 2>&1
tools/lab_prof.sh r3nt bench 3 > /dev/null 2>&1
tools/lab_prof.sh r3wg wbench 3 > /dev/null 2>&1
cut -c1-300 gpurun_out/r3_bench_line.json; tail -2 gpurun_out/pmc_r3.log; grep -c OK gpurun_out/r3_lab_check.log
