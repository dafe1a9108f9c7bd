#!/usr/bin/env python3
"""Which lines of the product launch the small torch kernels of a step?  (development tool)

torch.profiler with Python stacks over a few steady-state bench steps; every device kernel that is NOT one of libcoin_hip's is
charged to the innermost coin_amd/ frame of the op that launched it.

    python tools/glueprof.py [--steps 3] [--out gpurun_out/glueprof.txt]
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default="gpurun_out/glueprof.txt")
    args = ap.parse_args()
    import torch
    from torch.profiler import ProfilerActivity, profile

    import bench
    from coin_amd.engine import PRETrainer

    torch.backends.cudnn.benchmark = True
    cfg = bench.build_cfg(1, "cuda:0", "bf16")
    torch.manual_seed(cfg.SEED)
    tr = PRETrainer(cfg)
    for _ in range(5):
        tr.run_step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(args.steps):
            tr.run_step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.key_averages(group_by_stack_n=12):
        dt = getattr(e, "self_device_time_total", 0) or 0
        if dt <= 0 or not e.key.startswith("aten::"):
            continue
        frame = "?"
        for f in (e.stack or []):
            if "coin_amd/" in f and "kernels.py" not in f:
                frame = f[f.index("coin_amd/"):]
                break
        a = agg[(e.key, frame)]
        a[0] += e.count
        a[1] += dt
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    total = sum(v[1] for v in agg.values())
    lines = [f"aten ops with device time over {args.steps} steps: {total / args.steps / 1e3:.2f} ms/step"]
    for (name, frame), (n, t) in rows[:120]:
        lines.append(f"{t / args.steps:9.1f} us/step  n/step={n / args.steps:6.1f}  {name:28s} {frame}")
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:60]))


if __name__ == "__main__":
    main()
