mkdir -p gpurun_out/r5final2
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r5final2/pytest_gpu.log 2>&1; echo "full suite rc=$?"; grep -i -m3 "fault\|abort" gpurun_out/r5final2/pytest_gpu.log; tail -3 gpurun_out/r5final2/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/r5final2/smoke.log 2>&1; tail -1 gpurun_out/r5final2/smoke.log
bash tools/pmc_bench.sh r5final2 > gpurun_out/r5final2/pmc.log 2>&1; tail -2 gpurun_out/r5final2/pmc.log | cut -c1-200
bash tools/profile_bench.sh r5final2 > gpurun_out/r5final2/profile.log 2>&1; head -3 gpurun_out/prof_r5final2/steady_top.txt
timeout 900 python bench.py > gpurun_out/r5final2/bench.log 2>&1; tail -1 gpurun_out/r5final2/bench.log | cut -c1-250
