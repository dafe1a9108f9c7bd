#!/usr/bin/env python3
"""Which lines of the product call the small aten ops of a step, forward AND backward?  (development tool, GPU box)

torch.profiler with Python stacks over a few eager steady-state steps (COIN_STEP_GRAPHS=0 so that the captured stretches show their
ops); every aten op that launches a device kernel is charged to the innermost coin_amd/ (or bench / engine) frame of its stack;
backward ops (no Python frame of ours on the engine's thread) are charged to the autograd node that ran them.

    COIN_STEP_GRAPHS=0 python tools/opsites.py [--steps 2] [--out gpurun_out/opsites.txt] [--ops copy_,fill_,mul,add_,cat,add,to]
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--out", default="gpurun_out/opsites.txt")
    ap.add_argument("--targetdet", action="store_true", help="the CoinTrainer step (tools/bench_targetdet.py's trainer, 3 images) instead of the pre-train step")
    ap.add_argument("--step-two", action="store_true")
    args = ap.parse_args()
    import torch
    from torch.profiler import ProfilerActivity, profile

    import bench
    from coin_amd.engine import PRETrainer

    if args.targetdet:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_targetdet

        tr = bench_targetdet.build_trainer("foggy", 3, args.step_two)
        step = lambda: (tr.run_step(), tr.prepare_next())
    else:
        cfg = bench.build_cfg(1, "cuda:0", "bf16")
        torch.manual_seed(cfg.SEED)
        tr = PRETrainer(cfg)
        step = tr.run_step
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    try:
        xc = {"experimental_config": torch._C._profiler._ExperimentalConfig(verbose=True)}   # python stacks in events() on this torch
    except Exception:
        xc = {}
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True, **xc) as prof:
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
    evs = prof.events()
    # parent chains: an aten op launched inside an autograd node's evaluation has that node ("XBackward", "autograd::engine::evaluate_function: X") as an ancestor
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in evs:
        if not e.name.startswith("aten::"):
            continue
        dt = getattr(e, "self_device_time_total", 0) or 0
        if dt <= 0:
            continue
        site = None
        for f in (e.stack or []):
            if ("coin_amd/" in f or "bench.py" in f) and "kernels.py" not in f and "graphs.py" not in f:
                site = f[f.index("coin_amd/"):] if "coin_amd/" in f else f
                break
        if site is None:
            p = e.cpu_parent
            while p is not None:
                if "Backward" in p.name or "evaluate_function" in p.name:
                    site = "bwd: " + p.name
                    break
                p = p.cpu_parent
        shapes = str(getattr(e, "input_shapes", ""))[:60]
        if args.targetdet:
            shapes = ""
        a = agg[(e.name, site or "?", shapes)]
        a[0] += 1
        a[1] += dt
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    total = sum(v[1] for v in agg.values())
    lines = [f"aten ops with device time over {args.steps} steps: {total / args.steps / 1e3:.2f} ms/step"]
    for (name, site, shapes), (n, t) in rows[:150]:
        lines.append(f"{t / args.steps:9.1f} us/step  n/step={n / args.steps:6.1f}  {name:26s} {site}  {shapes}")
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:100]))


if __name__ == "__main__":
    main()
