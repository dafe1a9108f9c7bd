#!/usr/bin/env python3
"""Host wall time per phase of the targetDET step (development tool): where the Python thread spends a step.

    python tools/host_timeline.py [--config foggy|bdd100k_rn101] [--images 2] [--step-two]
"""
import argparse
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="foggy")
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--step-two", action="store_true")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=10)
    args = ap.parse_args()
    import torch

    import bench_targetdet as B

    tr = B.build_trainer(args.config, args.images, args.step_two)
    acc = collections.defaultdict(float)

    def timed(obj, name, label):
        fn = getattr(obj, name)

        def wrap(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                acc[label] += time.perf_counter() - t
        setattr(obj, name, wrap)

    t_model, s_model = tr.offline_teacher, tr.model
    timed(tr, "update_teacher", "teacher: EMA")
    timed(t_model.backbone, "forward", "teacher: backbone enqueue")
    timed(t_model.proposal_generator, "forward", "teacher: RPN + proposals (1 sync)")
    timed(t_model.roi_heads, "_pooled", "teacher: RoIAlign + res5 enqueue")
    timed(t_model.roi_heads.box_predictor, "inference", "teacher: per-image post-processing (syncs)")
    timed(tr, "match_boxes", "matching (host)")
    timed(tr, "_fetch", "_fetch total")
    timed(tr.optimizer, "step", "student: optimizer enqueue")
    timed(tr.optimizer_merge, "step", "CKG: optimizer enqueue")
    timed(tr, "run_step", "run_step total")
    real_call = type(s_model).__call__
    for _ in range(args.warmup):
        tr.run_step()
        tr.prepare_next()
    torch.cuda.synchronize()
    acc.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.run_step()
        tr.prepare_next()
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print(f"{args.config} images={args.images} step_two={args.step_two}: {total / args.steps * 1e3:.1f} ms/step")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"  {v / args.steps * 1e3:7.2f} ms/step  {k}")


if __name__ == "__main__":
    main()
