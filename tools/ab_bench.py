#!/usr/bin/env python3
"""Same-box A/B of the default bench with a switch of the LAB library flipped (GPU box; tools/build_lab.sh first):

    python tools/ab_bench.py no_s4 [rounds] [-- bench args]

Runs `bench.py --no-cpu-baseline --no-secondary` alternately with the switch off / on, each in its own process (the lab library is
loaded through coin_amd._lib.LIB_PATH, the switch is set before the first kernel), and prints ms_per_step per run + the medians.
Boxes differ by several percent in clock: only a comparison taken on one box, interleaved, says what a change buys.
"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    which, val = sys.argv[2], int(sys.argv[3])
    sys.path.insert(0, ROOT)
    from coin_amd import _lib

    _lib.LIB_PATH = os.path.join(ROOT, "tools", "lab", "libcoin_hip_lab.so")
    getattr(_lib.lib(), "coin_lab_set_" + which)(val)
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-secondary"] + sys.argv[4:]
    import runpy

    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
    sys.exit(0)

which = sys.argv[1] if len(sys.argv) > 1 else "no_s4"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
res = {0: [], 1: []}
for r in range(rounds):
    for val in (1, 0):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", which, str(val)] + extra, capture_output=True, text=True, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(out.stdout[-2000:], out.stderr[-2000:])
            sys.exit(1)
        d = json.loads(line[-1])
        res[val].append(d["ms_per_step"])
        print(json.dumps({"switch": which, "value": val, "ms_per_step": round(d["ms_per_step"], 3), "views_per_s": round(d["value"], 2),
                          "final_loss": d.get("config", {}).get("final_loss")}), flush=True)
print(json.dumps({"switch": which, "median_ms_off": statistics.median(res[0]), "median_ms_on": statistics.median(res[1]),
                  "on_minus_off_ms": statistics.median(res[1]) - statistics.median(res[0])}))
