#!/usr/bin/env python3
"""cProfile of the targetDET step's HOST side (development tool): where does the Python thread spend an iteration?

    python tools/prof_host_targetdet.py [--step-two] [--images 2] [--steps 12]
"""
import argparse
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--step-two", action="store_true")
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--top", type=int, default=45)
    args = ap.parse_args()
    import torch

    from bench_targetdet import build_trainer

    tr = build_trainer("foggy", args.images, args.step_two)
    for _ in range(10):
        tr.run_step()
        tr.prepare_next()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        tr.run_step()
        tr.prepare_next()
    torch.cuda.synchronize()
    pr.disable()
    for key in ("cumulative", "tottime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(args.top)
        print(f"==== sorted by {key} ({args.steps} iterations)")
        print("\n".join(l[:190] for l in s.getvalue().splitlines()[4:]))


if __name__ == "__main__":
    main()
