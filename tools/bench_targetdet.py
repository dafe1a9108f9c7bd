#!/usr/bin/env python3
"""Timing of the targetDET distillation step (BASELINE.json configs[2] shape) on ONE GPU -- informational, not the bench.py metric.

    python tools/bench_targetdet.py [--steps 8] [--warmup 4] [--images 2] [--step-two] [--sync-free-step]
    python tools/bench_targetdet.py --config bdd100k_rn101 --images 8          # BASELINE configs[3]: RN101, 750x1333, 8 images / GPU

Reports ms/step and student views/s (a step = teacher inference on the weak views + matching + student step on the strong views).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_trainer(config: str, images: int, step_two: bool, reference_samplers: bool = False, teacher_stream: bool = True, extra=()):
    """CoinTrainer of one of the synthetic targetDET configs; the (random-init) teacher's detections are replaced by CLIPDET-like ones
    AFTER its real inference pass has run, so that the A / B / C sets are populated."""
    import torch

    from coin_amd.config import get_cfg
    from coin_amd.data.synthetic import synthetic_offline_detections
    from coin_amd.engine import CoinTrainer

    torch.backends.cudnn.benchmark = True
    cfg = get_cfg()
    files = {"foggy": ("GDINO", "foggy_synthetic.yaml"), "bdd100k_rn101": ("GDINO", "bdd100k_rn101_synthetic.yaml"),
             "swint_fpn": ("FPN", "targetdet_swint_fpn_synthetic.yaml"), "rn101_fpn": ("FPN", "targetdet_rn101_fpn_synthetic.yaml")}
    cfg.merge_from_file(os.path.join(ROOT, "configs", "coin", *files[config]))
    cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", images, "AMD.SYNTHETIC.NUM_IMAGES", images, "AMD.TEXT_TEMPLATES", 4,
                         "MODEL.DEVICE", "cuda:0", "CLOUD.BURN_UP_STEP", 0 if step_two else 10 ** 9, "CLOUD.PROTOTYPE_UPDATE_START", 0,
                         "AMD.SYNC_FREE_STEP", not reference_samplers, "AMD.TEACHER_STREAM", teacher_stream] + list(extra))
    torch.manual_seed(cfg.SEED)
    tr = CoinTrainer(cfg)
    real_forward, g = tr.offline_teacher.forward, torch.Generator().manual_seed(7)

    def teacher(batched_inputs, branch=None, **kw):  # real teacher inference is executed and timed; its (random-init) detections
        real_forward(batched_inputs, branch=branch, **kw)  # are replaced by CLIPDET-like ones so that A/B/C sets are populated
        return [synthetic_offline_detections(tr.model_CLOUD.entry(d["file_name"]), g, device="cuda:0") for d in batched_inputs]

    tr.offline_teacher.forward = teacher
    tr.max_iter = 10 ** 9
    return tr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--step-two", action="store_true")
    ap.add_argument("--config", default="foggy", choices=["foggy", "bdd100k_rn101", "swint_fpn", "rn101_fpn"],
                    help="swint_fpn / rn101_fpn: the FPN extension (configs/coin/FPN, no counterpart in the reference)")
    ap.add_argument("--sync-free-step", action="store_true", help="(default since round 2) cfg.AMD.SYNC_FREE_STEP: fixed-shape samplers + packed losses")
    ap.add_argument("--reference-samplers", action="store_true", help="cfg.AMD.SYNC_FREE_STEP off: the reference-shaped nonzero / randperm samplers")
    ap.add_argument("--no-prefetch", action="store_true", help="cfg.AMD.TEACHER_PREFETCH off: the frozen teacher's pass follows the student's step (A/B measurement)")
    ap.add_argument("--after-pretrain", type=int, default=0, help="first run this many PRETrainer steps in the same process (the round-5 64-vs-52 ms mode experiment)")
    ap.add_argument("--no-teacher-stream", action="store_true", help="cfg.AMD.TEACHER_STREAM off: teacher pass on the main stream (A/B measurement)")
    args = ap.parse_args()
    import torch

    if os.environ.get("PRE_STREAMS"):   # experiment: side streams created (and used once) before the trainer exists, as a trainer that lived in the process earlier leaves them
        keep = [torch.cuda.Stream() for _ in range(int(os.environ["PRE_STREAMS"]))]
        for st in keep:
            with torch.cuda.stream(st):
                torch.zeros(1024, device="cuda").add_(1)
        torch.cuda.synchronize()
    if args.after_pretrain:   # what bench.py's process looked like in round 5: a PRETrainer has lived (and used its streams) here before
        import bench
        from coin_amd.engine import PRETrainer

        pcfg = bench.build_cfg(1, "cuda:0", "bf16")
        torch.manual_seed(pcfg.SEED)
        ptr = PRETrainer(pcfg)
        for _ in range(args.after_pretrain):
            ptr.run_step()
        torch.cuda.synchronize()
        del ptr
        torch.cuda.empty_cache()
    tr = build_trainer(args.config, args.images, args.step_two, reference_samplers=args.reference_samplers, teacher_stream=not args.no_teacher_stream,
                       extra=["AMD.TEACHER_PREFETCH", not args.no_prefetch])
    for _ in range(args.warmup):
        tr.run_step()
        tr.prepare_next()   # as CoinTrainer.train(): the next iteration's teacher pass / matching overlaps this backward
    torch.cuda.synchronize()
    # timed in groups of 4 steps (a device synchronize only between groups, so the pipelining inside a group is undisturbed): the
    # mean is what a long run sees; the median group is reported too because a RoI count that MIOpen has not met before costs a
    # kernel search of tens of milliseconds in the step where it first appears
    groups = []
    t0 = time.perf_counter()
    done = 0
    while done < args.steps:
        n = min(4, args.steps - done)
        tg = time.perf_counter()
        for _ in range(n):
            rec = tr.run_step()
            tr.prepare_next()
        torch.cuda.synchronize()
        groups.append((time.perf_counter() - tg) / n * 1e3)
        done += n
    dt = (time.perf_counter() - t0) / args.steps
    in_order = [round(g, 1) for g in groups]
    groups.sort()
    print(json.dumps({"config": args.config, "workload": "targetDET " + ("step_two" if args.step_two else "step_one") + (" (reference-shaped samplers)" if args.reference_samplers else "") + (" (teacher on the main stream)" if args.no_teacher_stream else "") + (" (no teacher prefetch)" if args.no_prefetch else ""), "images_per_step": args.images, "ms_per_step": dt * 1e3, "median_group_ms_per_step": groups[len(groups) // 2], "fastest_group_ms_per_step": groups[0], "groups_ms_per_step_in_order": in_order,
                      "student_views_per_s": args.images / dt, "losses": {k: round(float(v), 4) for k, v in rec.items()},
                      "final_loss": float(sum(float(v) for v in rec.values())), "steps": args.steps, "warmup": args.warmup,
                      "step_graphs": dict(__import__("coin_amd.graphs", fromlist=["STATS"]).STATS)}))


if __name__ == "__main__":
    main()
