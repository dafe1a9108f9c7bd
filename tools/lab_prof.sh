#!/bin/bash
# rocprofv3 passes over tools/gemm_lab (run on the GPU box from the repo root):  tools/lab_prof.sh <tag> <lab mode> [iters]
# pass 1: kernel trace + stats; passes 2-4: PMC (SQ set, FETCH_SIZE, WRITE_SIZE), kernel-trace only, one counter set per pass.
tag=$1; mode=$2; iters=${3:-3}
out=$PWD/gpurun_out/labprof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $OLDPWD/tools/gemm_lab $mode $iters > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/sq -- $OLDPWD/tools/gemm_lab $mode $iters > $out/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $OLDPWD/tools/gemm_lab $mode $iters > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/write -- $OLDPWD/tools/gemm_lab $mode $iters > $out/write.log 2>&1
cd $OLDPWD
python3 tools/lab_prof_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
