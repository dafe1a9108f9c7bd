set -x
mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "pooled_residual or fan_in or conv_gemm" > gpurun_out/r5a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5a/pytest.log
COIN_STEP_GRAPHS=0 timeout 600 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r5a/b_off.json 2> gpurun_out/r5a/b_off.err; echo "rc=$?"
COIN_STEP_GRAPHS=1 timeout 600 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r5a/b_on.json 2> gpurun_out/r5a/b_on.err; echo "rc=$?"
tail -3 gpurun_out/r5a/pytest.log
tail -5 gpurun_out/r5a/b_on.err
python - <<'PY'
import json
for n in ("off","on"):
    try:
        d=json.loads(open(f"gpurun_out/r5a/b_{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["value"],2), round(d["ms_per_step"],2), d["config"].get("host_enqueue_ms"), d["config"].get("step_graphs"), "roof", d["roofline"] and round(d["roofline"]["frac"],4), "sec", d.get("secondary",{}).get("value"), d.get("secondary",{}).get("groups_ms_per_step_in_order"), d.get("secondary",{}).get("step_graphs"), d.get("secondary",{}).get("error"))
    except Exception as e:
        print(n, "ERR", e)
PY
