#!/bin/bash
# Round-6 evidence set (GPU box): bench line, rocprofv3 kernel stats + steady state, PMC traffic, GEMM lab, targetDET account.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r6_ev
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $out/bench_line.log 2>&1
tail -1 $out/bench_line.log > $out/bench_line.json
bash tools/profile_bench.sh r6 > /dev/null 2>&1
bash tools/pmc_bench.sh r6 > $out/pmc.log 2>&1
python tools/pmc_traffic_json.py gpurun_out/prof_r6/pmc_bench_FETCH_SIZE.csv gpurun_out/prof_r6/pmc_bench_WRITE_SIZE.csv $out/bench_line.json $out/pmc_traffic.json > $out/pmc_traffic.log 2>&1
( echo "# tools/gemm_lab scheck + swcheck + check + wcheck (COIN_LAB build), round 6"; timeout 600 tools/gemm_lab scheck; timeout 600 tools/gemm_lab swcheck; timeout 900 tools/gemm_lab check; timeout 600 tools/gemm_lab wcheck ) > $out/lab_check.log 2>&1
( echo "# tools/gemm_lab sbench 30 / wbench 30 (COIN_LAB build), round 6: hot caches, us per launch"; LAB_SHAPES="l3.x.conv1 l3.x.conv2 l3.x.conv3 l2.x.conv1 l2.x.conv2 l2.x.conv3 l2.0.conv1 l2.0.conv2 l2.0.down l3.0.conv1 l3.0.conv2 l3.0.down rpn.conv l4.0.conv3 l4.1.conv2" timeout 600 tools/gemm_lab sbench 30; LAB_SHAPES="l3.x.conv1 l3.x.conv2 l3.x.conv3 l2.x.conv1 l2.x.conv2 l2.x.conv3 l2.0.conv1 l2.0.conv2 l2.0.down l3.0.conv1 l3.0.conv2 l3.0.down head.fc1 head.fc2 rpn.conv l4.0.conv1 l4.0.conv2 l4.0.conv3 l4.0.down l4.1.conv1 l4.1.conv2" timeout 600 tools/gemm_lab wbench 30 ) > $out/lab_bench.log 2>&1
bash tools/profile_bench.sh r6td1 tools/bench_targetdet.py --images 3 --steps 16 --warmup 8 > /dev/null 2>&1
bash tools/profile_bench.sh r6td2 tools/bench_targetdet.py --images 3 --steps 16 --warmup 8 --step-two > /dev/null 2>&1
for i in 1 2; do timeout 300 python tools/bench_targetdet.py --images 3 --steps 24 --warmup 12 | tail -1; timeout 300 python tools/bench_targetdet.py --images 3 --steps 24 --warmup 12 --step-two | tail -1; done > $out/targetdet.jsonl 2>/dev/null
( for c in bdd100k_rn101 swint_fpn rn101_fpn; do timeout 400 python tools/bench_targetdet.py --config $c --images $([ $c = bdd100k_rn101 ] && echo 8 || echo 3) --steps 16 --warmup 8 | tail -1; done ) >> $out/targetdet.jsonl 2>/dev/null
ls -la $out
