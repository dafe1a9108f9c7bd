#!/usr/bin/env python3
"""Where does a training step spend its time?  (development tool, not part of the product)

1. host- vs device-bound: wall time for the host to ENQUEUE K steps vs. wall time until the device finished them;
2. torch.profiler table of aten/custom ops grouped by input shape, sorted by device time, for a few steady-state steps.

    python tools/opprof.py [--steps 4] [--out gpurun_out/opprof.txt]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--out", default="gpurun_out/opprof.txt")
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    import torch

    import bench
    from coin_amd.engine import PRETrainer

    torch.backends.cudnn.benchmark = True
    if os.environ.get("COIN_FORCE_DDP") == "1":  # profile the DDP path with a single rank
        import torch.distributed as dist

        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29512"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    cfg = bench.build_cfg(1, "cuda:0", args.dtype)
    torch.manual_seed(cfg.SEED)
    tr = PRETrainer(cfg)
    for _ in range(4):
        tr.run_step()
    torch.cuda.synchronize()
    lines = []
    # --- 1. enqueue vs complete
    t0 = time.perf_counter()
    for _ in range(10):
        tr.run_step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    lines.append(f"10 steps: host enqueue {t_enq * 100:.2f} ms/step, complete {t_all * 100:.2f} ms/step "
                 f"({'HOST' if t_enq > 0.9 * t_all else 'DEVICE'}-bound)")
    # phases of one step on the host (synchronising between phases: device time per phase)
    def phase(fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        t_h = time.perf_counter() - t
        torch.cuda.synchronize()
        return r, t_h * 1e3, (time.perf_counter() - t) * 1e3

    for _ in range(2):
        strong, weak = next(tr._data_loader_iter)
        strong, weak = tr.set_boxes([strong, weak])
        strong.extend(weak)
        rec, h1, d1 = phase(lambda: tr.model(strong, branch="pre_train", update_prototype=False))
        loss = sum(rec.values())
        _, h0, d0 = phase(lambda: tr.optimizer.zero_grad())
        _, h2, d2 = phase(lambda: loss.backward())
        _, h3, d3 = phase(lambda: tr.optimizer.step())
        lines.append(f"phases (host ms / until-idle ms): forward {h1:.1f}/{d1:.1f}  zero_grad {h0:.1f}/{d0:.1f}  backward {h2:.1f}/{d2:.1f}  sgd {h3:.1f}/{d3:.1f}")
    # --- 2. op table
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(args.steps):
            tr.run_step()
        torch.cuda.synchronize()
    ka = prof.key_averages(group_by_input_shape=True)
    lines.append(ka.table(sort_by="self_cuda_time_total", row_limit=90, max_name_column_width=60, max_shapes_column_width=90))
    lines.append(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))
    # elementwise glue by (op, input shapes): what the fused kernels do not cover yet
    glue = [e for e in ka if e.key.startswith("aten::") and not any(t in e.key for t in ("conv", "mm", "linear", "matmul", "attention", "layer_norm"))]
    dev_t = lambda e: getattr(e, "self_device_time_total", None) if hasattr(e, "self_device_time_total") else e.self_cuda_time_total
    glue.sort(key=lambda e: -dev_t(e))
    lines.append("\nelementwise / copy glue by input shape (self device time over %d steps):" % args.steps)
    for e in glue[:60]:
        if dev_t(e) <= 0:
            break
        lines.append(f"{dev_t(e) / 1e3 / args.steps:8.3f} ms/step  n/step={e.count / args.steps:6.1f}  {e.key:28s} {str(e.input_shapes)[:150]}")
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        f.write("\n".join(lines))
    print("\n".join(lines[:3]))


if __name__ == "__main__":
    main()
