mkdir -p gpurun_out/r5c
td() { name=$1; shift; env "$@" timeout 600 python tools/bench_targetdet.py $ARGS > gpurun_out/r5c/td_$name.log 2>&1; echo "$name rc=$? $(grep -o '"ms_per_step": [0-9.]*\|"groups_ms_per_step_in_order": \[[^]]*\]\|"student_views_per_s": [0-9.]*' gpurun_out/r5c/td_$name.log | tr '\n' ' ')"; grep -i "warn.*graph\|capture failed" gpurun_out/r5c/td_$name.log | head -3; }
ARGS="--images 3"
td one_off COIN_STEP_GRAPHS=0
td one_on COIN_STEP_GRAPHS=1
td one_on_tg COIN_STEP_GRAPHS=1 COIN_TEACHER_GRAPH=always
ARGS="--images 2 --step-two"
td two_off COIN_STEP_GRAPHS=0
td two_on COIN_STEP_GRAPHS=1
timeout 900 tools/gemm_lab check > gpurun_out/r5c/lab_check.log 2>&1; grep -c OK gpurun_out/r5c/lab_check.log; grep -v " OK" gpurun_out/r5c/lab_check.log | tail -8
LAB_SHAPES="l2.x.conv2,l2.x.conv1,l3.x.conv2" timeout 600 tools/gemm_lab bench 10 > gpurun_out/r5c/lab_bench_l2.log 2>&1; tail -8 gpurun_out/r5c/lab_bench_l2.log | cut -c1-400
