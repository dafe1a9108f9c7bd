#!/usr/bin/env python3
"""Benchmark of COIN's adaptation-training hot path on MI355X (contract: see the task prompt / DESIGN.md §4).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N --steps K --warmup W          # starts its own N ranks (one process per GPU, RCCL over 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # ... or runs as one rank of an outer launcher (RANK / WORLD_SIZE set)

One "step" = one ``PRETrainer.run_step`` of the CLIPDET pre-training config (BASELINE.json configs[1]):
per GPU 2 synthetic 800x1333 VOC-shaped images x (strong + weak view) = 4 views, each forward + backward
through CLIP-RN50-C4 backbone, RPN, RoIAlign, res5 on 512 sampled RoIs, box predictor, losses, then SGD.
Inputs (uint8 views, cached teacher boxes) are resident in HBM before the timed region.
``value`` = views ("images") per second over all GPUs (weak scaling: 4 views per GPU per step).

Rank 0 prints ONE JSON line with ``roofline`` (dominant hand-written kernel, timed live with HIP events on its
launch stream) and ``cpu_baseline`` (the CPU oracle's training step on the host cores, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the HIP runtime starts: see coin_amd/__init__.py
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VIEWS_PER_IMAGE = 2          # strong + weak (pre_train.py:191)
IMAGES_PER_GPU = 2           # BASELINE.json configs[0]: "2 synthetic 800x1333 VOC images"
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0
FLOP_PER_VIEW = 4.259e12     # fwd 1449 + bwd 2811 GFLOP per 800x1333 view, RN50-C4, 512 RoIs


# C-ABI entry point -> the device kernels one call launches (rocprofv3 lists these; their average durations add up to the
# entry point's `mean_launch_ms`)
ENTRY_KERNELS = {
    "coin_roi_align_fwd": ["roi_align_fwd_cols_kernel  (pooled widths 7 / 14; roi_align_fwd_nhwc_kernel for other bin counts)"], "coin_roi_align_bwd": ["roi_align_bwd_gather_kernel"],
    "coin_bn_stats": ["bn_stats_kernel", "bn_finalize_kernel"], "coin_bn_apply_fwd": ["bn_apply_kernel | bn_apply_mean_kernel"],
    "coin_bn_bwd": ["bn_bwd_reduce_kernel", "bn_bwd_finalize_kernel", "bn_bwd_dx_kernel"], "coin_gemm_nt": ["gemm_nt_bf16_kernel"],
    "coin_conv_gemm_bf16": ["conv_gemm_p8_kernel (+ conv_gemm_p8_slab_sum_kernel, conv_gemm_p8_tail_kernel where leftover tiles are cut along K) | "
                            "conv_gemm_s4_kernel (round 6: the 128 x 128 small-map core)  (<GATHER3, STATS> instantiations)"],
    "coin_conv_wgrad_bf16": ["conv_wgrad_p8_kernel + tn_reduce_kernel | conv_wgrad_s4_kernel + conv_wgrad_s4_reduce_kernel (round 6: launches below 160 MB of operands)"],
    "coin_conv_gemm_stats_finalize": ["conv_stats_finalize_kernel"],
}
KERNEL_TIMING_STEPS = 2    # the first steps of the timed region carry HIP events around every timed launch (`kernels` / `roofline` blocks)
MFMA_ENTRIES = ("coin_gemm_nt", "coin_conv_gemm_bf16", "coin_conv_wgrad_bf16")   # their `units` slot carries FLOPs, not bytes


def pmc_traffic(entry: str, alg_bytes: float):
    """(HBM bytes per launch, the file it comes from) from the committed PMC passes (profiles/r6_pmc_traffic.json, else older rounds':
    FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KB units, MI355X_MICROARCH.md) -- only when that pass measured this entry point at this launch
    size (for the MFMA entry points: at this mean FLOP count per launch), else (None, None).  rocprofv3 cannot attach to a running
    process, so the figure is NOT measured by the run that prints the line: `roofline.traffic_source` names the file."""
    for name in ("r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                table = json.load(f)
            rec = table.get(entry)
            if rec and abs(rec["alg_bytes"] - alg_bytes) <= 0.01 * alg_bytes:
                return rec["hbm_bytes"], "profiles/" + name
        except (OSError, ValueError, KeyError):
            pass
    return None, None


def build_cfg(world: int, device: str, dtype: str, extra=()):
    from coin_amd.config import get_cfg

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "coin", "PRETRAINS", "CLIPDET_synthetic.yaml"))
    cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", IMAGES_PER_GPU * world, "MODEL.DEVICE", device, "AMD.COMPUTE_DTYPE", dtype,
                         "AMD.SYNTHETIC.NUM_IMAGES", IMAGES_PER_GPU * world, "AMD.TEXT_TEMPLATES", 4] + list(extra))
    return cfg


def cpu_baseline_main(args):
    """Child process: the oracle's CLIPDET pre-training step (fwd + bwd + SGD) on the host cores, ONE 800x1333 view."""
    import torch

    from oracle import coin as OC
    from oracle import d2

    ncpu = os.cpu_count() or 1
    torch.manual_seed(2024)
    # torch's CPU kernels do not scale to every core of a many-socket host (256 threads ran this step ~10x slower than 32):
    # pick the thread count that runs a res5-shaped conv block fastest, and report THAT as `cores`.
    probe = torch.nn.Sequential(torch.nn.Conv2d(512, 512, 3, padding=1, bias=False), torch.nn.BatchNorm2d(512), torch.nn.ReLU())
    xp = torch.randn(256, 512, 14, 14)  # a quarter of one view's res5 batch: large enough that the ranking carries over
    best, cores = None, 1
    for t in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(t)
        probe(xp).sum().backward()
        t0 = time.perf_counter()
        for _ in range(3):
            probe(xp).sum().backward()
        dt_p = time.perf_counter() - t0
        if best is None or dt_p < best:
            best, cores = dt_p, t
    torch.set_num_threads(cores)
    h, w = args.cpu_h, args.cpu_w
    model = OC.build_detector(num_classes=8, roi_batch=512, zero_init_bn3=True)
    model.train()
    groups = OC.optimizer_param_groups(model, 0.001, {"backbone.encoder.visual": 0.1}, weight_decay_norm=0.0, weight_decay_bias=1e-4)
    opt = torch.optim.SGD(groups, lr=0.001, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(2024)
    from coin_amd.data.synthetic import synthetic_teacher_result

    res = synthetic_teacher_result("cpu.png", "cpu", h, w, 32, 8, g)
    inst = res["RCNN"]["instances"]
    probs = inst.probs
    rc = d2.Instances((h, w))
    rc.gt_boxes = d2.Boxes(inst.pred_boxes.tensor.clone())
    rc.gt_classes_offline, rc.gt_probs_offline, rc.gt_scores_offline = inst.pred_classes, probs, inst.scores
    rp = d2.Instances((h, w))
    rp.gt_boxes = d2.Boxes(inst.pred_boxes.tensor.clone())
    rp.gt_classes = inst.pred_classes
    batch = [{"image": torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8), "RCNN": rc, "RPN": rp, "height": h, "width": w}]
    def make_step(mdl, optim):
        def step():
            losses = mdl(batch, branch="pre_train", update_prototype=False)
            total = sum(losses.values())
            optim.zero_grad()
            total.backward()
            optim.step()
            return float(total)
        return step

    step = make_step(model, opt)
    t0 = time.perf_counter()
    step()                      # warm-up: oneDNN primitive creation, allocator growth
    warm = time.perf_counter() - t0
    times = []
    for _ in range(args.cpu_steps):
        t0 = time.perf_counter()
        loss = step()
        times.append(time.perf_counter() - t0)
    out = {"seconds": sum(times) / len(times), "step_seconds": times, "warmup_seconds": warm, "cores": cores, "views": 1, "loss": loss,
           "nproc": ncpu, "cpu_model": _cpu_model()}
    print(json.dumps(out), flush=True)   # the multi-thread figure is safe even if the single-thread leg below is cut off
    # single-thread leg, bounded: the same view with 64 RoIs (a full 512-RoI step takes minutes on one core)
    torch.set_num_threads(1)
    small = OC.build_detector(num_classes=8, roi_batch=64, zero_init_bn3=True)
    small.train()
    sgroups = OC.optimizer_param_groups(small, 0.001, {"backbone.encoder.visual": 0.1}, weight_decay_norm=0.0, weight_decay_bias=1e-4)
    sstep = make_step(small, torch.optim.SGD(sgroups, lr=0.001, momentum=0.9, weight_decay=1e-4))
    t0 = time.perf_counter()
    sstep()
    out["single_thread"] = {"seconds": time.perf_counter() - t0, "views": 1, "rois": 64, "threads": 1}
    print(json.dumps(out), flush=True)


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run_cpu_baseline(timeout_s: int):
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"]
    stdout = ""
    try:
        try:
            # the CPU leg chooses its own thread count by a probe: it must not inherit this rank's thread caps / core set
            env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "COIN_RANK_CPUSET")}
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT, env=env)
            stdout = out.stdout
        except subprocess.TimeoutExpired as te:  # the multi-thread line is printed first: keep it if only the single-thread leg ran out of time
            stdout = te.stdout.decode() if isinstance(te.stdout, bytes) else (te.stdout or "")
        line = [l for l in stdout.strip().splitlines() if l.startswith("{")][-1]
        r = json.loads(line)
        res = {"value": r["views"] / r["seconds"], "unit": "images/sec", "cores": r["cores"], "kind": "port",
               "sample": "oracle/coin.py CLIPDET pre-train step (fwd+bwd+SGD, fp32, torch CPU) on ONE 800x1333 view with 512 RoIs: "
                         f"1 warm-up step ({r.get('warmup_seconds', 0.0):.1f} s) + {len(r.get('step_seconds', [0]))} timed steps "
                         f"({', '.join(f'{t:.1f}' for t in r.get('step_seconds', [r['seconds']]))} s)",
               "nproc": r.get("nproc"), "cpu_model": r.get("cpu_model")}
        st = r.get("single_thread")
        res["single_thread"] = ({"value": st["views"] / st["seconds"], "unit": "images/sec", "cores": 1,
                                 "sample": f"the same step on ONE 800x1333 view with {st['rois']} RoIs (bounded sample), 1 step, {st['seconds']:.1f} s"}
                                if st else {"value": None, "sample": "not measured within the time limit"})
        return res
    except Exception as e:  # timeout or failure: report it, never fake a number
        return {"value": None, "unit": "images/sec", "cores": os.cpu_count(), "kind": "port", "sample": f"not measured: {type(e).__name__}: {e}"[:300]}


def child_env(rank: int, world: int, port: int, base=None, allowed=None, topology=None) -> dict:
    """Environment of rank `rank` of a self-launched run (what torch.distributed.run would set, rendezvous on 127.0.0.1)."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    # host hygiene (coin_amd/hostenv.py): a disjoint NUMA-local core set per rank + OMP / MKL thread caps, decided by the launcher so that
    # all ranks agree; the rank pins itself (apply_rank_affinity) before the HIP runtime starts.  No re-exec anywhere.
    from coin_amd.hostenv import rank_env

    env.update(rank_env(rank, world, base=env, allowed=allowed, topology=topology))
    return env


def child_argv(argv) -> list:
    return [sys.executable, os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` without an outer launcher: this process starts N fresh rank processes and waits for them.  It never
    imports torch.cuda or touches a GPU itself (the ranks are children, not exec replacements: train_net.py:132-139 of the reference
    plays the same role with detectron2's launch())."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [subprocess.Popen(child_argv(argv), env=child_env(r, args.gpus, port), cwd=ROOT) for r in range(args.gpus)]
    # poll: a rank that dies (import error, out of memory) leaves its siblings blocked in the rendezvous / an all-reduce until the
    # process-group timeout -- end them (they are this process's own children) and report the first failure instead
    rc, live = 0, list(procs)
    while live and rc == 0:
        time.sleep(0.2)
        for pr in list(live):
            code = pr.poll()
            if code is not None:
                live.remove(pr)
                if code != 0:
                    rc = abs(code)
    for pr in live:
        pr.terminate()
    for pr in live:
        try:
            pr.wait(timeout=20)
        except subprocess.TimeoutExpired:
            pr.kill()
            pr.wait()
    return rc


def run_secondary_targetdet(steps: int = 24, warmup: int = 12, images: int = 3, timeout: float = 600.0):
    """`secondary` block of the bench line: BASELINE.json configs[2] (targetDET distillation, CLIP-RN50 C4 student + EMA teacher,
    Foggy-Cityscapes-shaped 667x1333 views) on this one GPU -- `CoinTrainer.run_step` + `prepare_next` exactly as `CoinTrainer.train()`
    issues them: teacher inference on the weak views, A/B/C matching against the cached cloud boxes, student step on the strong views.
    step_one is the block's `value`; round 6 adds a `step_two` sub-block (EMA teacher updated every step, the C-box pass).
    tools/bench_targetdet.py (groups of 4 steps, a device synchronize between groups only) in a CHILD PROCESS of its own, as the reference
    runs the two trainings as two jobs.  (Round 5 measured 64 instead of 53 ms for this step inside the benchmark's own process; round 6
    reproduces a 57-59 ms floor there against 47-48 ms in a process of its own -- `tools/bench_targetdet.py --after-pretrain 8`;
    coin_amd/streams.py lists what was tried -- so the child process stays: a job of its own is also what a user runs.)"""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "COIN_RANK_CPUSET")}   # a job of its own

    def one(extra):
        cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_targetdet.py"), "--images", str(images), "--steps", str(steps), "--warmup", str(warmup)] + extra
        pr = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
        lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
        if pr.returncode != 0 or not lines:
            return None, f"tools/bench_targetdet.py {' '.join(extra)} exited with {pr.returncode}: {pr.stderr[-400:]}"
        return json.loads(lines[-1]), None

    try:
        d, err = one([])
        if d is None:
            return {"error": err}
        loss = float(d["final_loss"])
        out = {"metric": "targetDET step_one student images/sec (667x1333, 512 RoI/img, teacher pass + A/B/C matching included)", "value": d["student_views_per_s"],
               "unit": "images/sec", "n_gpus": 1, "steps": d["steps"], "warmup": d["warmup"], "ms_per_step": d["ms_per_step"],
               "median_group_ms_per_step": d["median_group_ms_per_step"], "fastest_group_ms_per_step": d["fastest_group_ms_per_step"],
               "groups_ms_per_step_in_order": d["groups_ms_per_step_in_order"],   # a busy host shows as unequal groups
               "images_per_step": images, "dtype": "bf16", "data": "synthetic", "final_loss": loss, "finite": loss == loss and abs(loss) < 1e6,
               "step_graphs": d.get("step_graphs"), "process": "child (tools/bench_targetdet.py)",
               "config": {"workload": "BASELINE configs[2]: CoinTrainer.run_step + prepare_next, CLIP-RN50 C4/res5 student and EMA teacher (frozen in step_one), "
                                      "3 synthetic Foggy-Cityscapes-shaped images per step, 1000 teacher RoIs + 512 student RoIs per image, 8 classes"}}
        d2, err2 = one(["--step-two"])
        if d2 is None:
            out["step_two"] = {"error": err2}
        else:
            l2 = float(d2["final_loss"])
            out["step_two"] = {"metric": "targetDET step_two student images/sec (EMA teacher updated every step, C-box pass)", "value": d2["student_views_per_s"],
                               "unit": "images/sec", "ms_per_step": d2["ms_per_step"], "median_group_ms_per_step": d2["median_group_ms_per_step"],
                               "groups_ms_per_step_in_order": d2["groups_ms_per_step_in_order"], "steps": d2["steps"], "warmup": d2["warmup"],
                               "final_loss": l2, "finite": l2 == l2 and abs(l2) < 1e6}
        return out
    except Exception as e:  # the headline stands on its own: report, do not fail the line
        return {"error": f"{type(e).__name__}: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` block (targetDET step_one, measured after the headline's timed region)")
    ap.add_argument("--secondary-steps", type=int, default=24)
    ap.add_argument("--cpu-timeout", type=int, default=300)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--cpu-h", type=int, default=800)
    ap.add_argument("--cpu-w", type=int, default=1333)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        return cpu_baseline_main(args)
    if args.gpus > 1 and "RANK" not in os.environ:   # no outer launcher: become the launcher (before anything touches the GPU)
        raise SystemExit(self_launch(args, sys.argv[1:]))

    from coin_amd.hostenv import apply_rank_affinity, cap_torch_threads

    affinity = apply_rank_affinity()   # before the HIP runtime starts; under torch.distributed.run (the driver's launcher) from LOCAL_RANK
    import torch
    import torch.distributed as dist

    cap_torch_threads(affinity)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("COIN_FORCE_DDP") == "1" or os.environ.get("COIN_INIT_PG") == "1":  # COIN_FORCE_DDP: exercise the RCCL path with a single rank
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from coin_amd import graphs as G
    from coin_amd import kernels as K
    from coin_amd.engine import PRETrainer

    torch.backends.cudnn.benchmark = True
    cfg = build_cfg(world, f"cuda:{local_rank}", args.dtype)
    torch.manual_seed(cfg.SEED + rank)
    trainer = PRETrainer(cfg)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.run_step()
    sync()
    timed = ["coin_roi_align_fwd", "coin_roi_align_bwd", "coin_gemm_nt", "coin_bn_stats", "coin_bn_apply_fwd", "coin_bn_bwd",
             "coin_conv_gemm_bf16", "coin_conv_wgrad_bf16"]
    K.timing_begin(timed)
    host_ms = []   # per step: wall time the host spent inside run_step (enqueue; close to ms_per_step = the rank is host-bound)
    from coin_amd.telemetry import GpuTelemetry

    telemetry = GpuTelemetry(local_rank)   # shader clock / board power / temperature from sysfs on a side thread, DURING the timed region
    telemetry.__enter__()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == KERNEL_TIMING_STEPS:   # 460 events per step cost the host 1.3 ms per step, and tens of thousands of live events more
            K.timing_pause()           # (60 fully instrumented steps: 36 -> 49 ms/step, measured): the rest of the region runs bare
        th = time.perf_counter()
        rec = trainer.run_step()
        host_ms.append((time.perf_counter() - th) * 1e3)
    sync()
    dt = time.perf_counter() - t0
    telemetry.__exit__()
    bare = sorted(host_ms[KERNEL_TIMING_STEPS:] or host_ms)
    host_enqueue_ms = bare[len(bare) // 2]
    ktimes = K.timing_end()
    loss = float(sum(rec.values()))
    # AFTER the timed region: the same call with the device idle at its start (a synchronize before every step), i.e. what the host needs
    # to enqueue a step when it never has to wait for the device -- free-running, `host_enqueue_ms` includes such waits (a graph replay
    # blocks while the previous launch of that graph is still running), so it reads like the step time on a device-bound rank
    idle_ms = []
    for _ in range(5):
        sync()
        th = time.perf_counter()
        trainer.run_step()
        idle_ms.append((time.perf_counter() - th) * 1e3)
    sync()
    host_enqueue_idle_ms = sorted(idle_ms)[len(idle_ms) // 2]
    per_rank = [[dt, loss, host_enqueue_ms]]
    affinities = [affinity]
    if world > 1:   # every rank's wall time, final loss and host enqueue time: the line reports the slowest rank (value) and the spread
        mine = torch.tensor([dt, loss, host_enqueue_ms], device="cuda", dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[float(a[0]), float(a[1]), float(a[2])] for a in allr]
        dt = max(r[0] for r in per_rank)
        affinities = [None] * world
        dist.all_gather_object(affinities, affinity)
    views_per_step = IMAGES_PER_GPU * VIEWS_PER_IMAGE * world
    value = views_per_step * args.steps / dt
    for r, (_t, lr_, _h) in enumerate(per_rank):
        if not (lr_ == lr_ and abs(lr_) < 1e6):
            raise SystemExit(f"bench.py: the training loss of rank {r} is not finite ({lr_}) after {args.warmup + args.steps} steps: the run is invalid")

    if rank == 0:
        # dominant hand-written kernel by total time inside the timed region
        best = max(ktimes.items(), key=lambda kv: kv[1][0] * kv[1][1]) if ktimes else None
        roofline = None
        detail = {}
        tsteps = min(args.steps, KERNEL_TIMING_STEPS)
        for name, (n, ms, units) in ktimes.items():
            if name in MFMA_ENTRIES:
                detail[name] = {"launches": n, "mean_ms": ms, "total_ms_per_step": n * ms / min(args.steps, KERNEL_TIMING_STEPS), "TFLOP/s": units / (ms * 1e-3) / 1e12,
                                "alg_flop": units, "device_kernels": ENTRY_KERNELS.get(name)}
                if name in K.SHAPES:   # the launch mix behind the mean: one row per (M, N, K, kernel size), by time
                    rows = sorted(K.SHAPES[name].items(), key=lambda kv: -kv[1][1])
                    detail[name]["shapes"] = [{"M": t[0], "N": t[1], "K": t[2], "ks": t[3], "calls_per_step": c / tsteps, "ms": round(tot / c, 4),
                                               "ms_per_step": round(tot / tsteps, 4), "TFLOP/s": round(fl / (tot / c * 1e-3) / 1e12, 1)} for t, (c, tot, fl) in rows]
            else:
                detail[name] = {"launches": n, "mean_ms": ms, "total_ms_per_step": n * ms / min(args.steps, KERNEL_TIMING_STEPS), "GB/s": units / (ms * 1e-3) / 1e9,
                                "alg_bytes": units, "device_kernels": ENTRY_KERNELS.get(name)}
        if best is not None:
            name, (n, ms, units) = best
            if name in MFMA_ENTRIES:
                ach = units / (ms * 1e-3) / 1e12
                traffic, tsrc = pmc_traffic(name, units)
                roofline = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic, "traffic_source": tsrc, "alg_flop_per_launch": units,
                            "device_kernels": ENTRY_KERNELS.get(name)}
            else:
                ach = units / (ms * 1e-3) / 1e9
                traffic, tsrc = pmc_traffic(name, units)
                roofline = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": tsrc, "alg_bytes_per_launch": units,
                            "device_kernels": ENTRY_KERNELS.get(name)}
            if name in K.SHAPES and name in MFMA_ENTRIES:
                # Since round 5 the launch mix of this entry point also holds the backbone's small maps (the convolutions of layer3 / layer2 that
                # ran on the library before: a captured stretch must not contain library launches).  The round-4 mix -- launches
                # of at least 32 768 rows + the long-K 3x3 exception -- is priced separately so that the figure stays comparable across rounds.
                big = [(c, tot, fl) for t, (c, tot, fl) in K.SHAPES[name].items() if t[0] >= 32768 or (t[3] == 3 and t[0] >= 16384 and t[1] % 256 == 0 and t[2] % 2304 == 0)]
                if big and sum(tot for _, tot, _ in big) > 0:
                    ach4 = sum(c * fl for c, _, fl in big) / (sum(tot for _, tot, _ in big) * 1e-3) / 1e12
                    roofline["round4_launch_mix"] = {"launches": sum(c for c, _, _ in big), "achieved": ach4, "frac": ach4 / MFMA_BF16_PEAK_TFLOPS,
                                                     "ms_per_step": sum(tot for _, tot, _ in big) / tsteps}
            roofline["launches_timed"] = n
            roofline["kernel_timing_steps"] = min(args.steps, KERNEL_TIMING_STEPS)
            roofline["mean_launch_ms"] = ms
        out = {
            "metric": "adaptation-train images/sec (800x1333, 512 RoI/img)", "value": value, "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "CLIPDET pre-train step (PRETrainer.run_step): CLIP-RN50 C4/res5 detector, per GPU 2 synthetic 800x1333 "
                                   "images x (strong+weak) = 4 views, real RPN + sampler, 512 RoIs/view, 8 classes, 32 cached teacher "
                                   "boxes/image, random-init weights, SGD",
                       "views_per_gpu_per_step": IMAGES_PER_GPU * VIEWS_PER_IMAGE, "parallelism": f"dp{world}", "final_loss": loss,
                       "rank_ms_per_step": {"min": min(r[0] for r in per_rank) / args.steps * 1e3, "max": max(r[0] for r in per_rank) / args.steps * 1e3},
                       "rank_final_loss": [r[1] for r in per_rank],
                       # median wall time per step the host spent enqueueing (run_step call to return, un-instrumented steps): a rank whose
                       # figure is close to ms_per_step is host-bound, one well below it is device-bound
                       "host_enqueue_ms": [round(r[2], 2) for r in per_rank],
                       "host_enqueue_device_idle_ms_rank0": round(host_enqueue_idle_ms, 2),   # 5 extra steps after the timed region, see above
                       "rank_affinity": affinities,
                       # what the chip held during the timed region (sysfs, rank 0's GPU): boxes of one pool differ by several percent in
                       # clock under these kernels -- a slow box and a regression look the same without it
                       "gpu_telemetry": telemetry.summary(),
                       # coin_amd/graphs.py: captured (shape, segment) pairs, graph replays / eager calls of the two graphed stretches so far
                       "step_graphs": dict(G.STATS, enabled=G.ENABLED["on"]),
                       "end_to_end_mfma_frac": value / world * FLOP_PER_VIEW / (MFMA_BF16_PEAK_TFLOPS * 1e12)},
            "roofline": roofline, "kernels": detail,
        }
        if world == 1 and not args.no_secondary:
            # AFTER the headline's timed region (which is untouched by it): one driver-timed number for the widened row (f)-1
            del trainer
            torch.cuda.empty_cache()
            out["secondary"] = run_secondary_targetdet(args.secondary_steps)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = run_cpu_baseline(args.cpu_timeout)
    else:
        out = None
    # The JSON line must be the LAST line on stdout: RCCL writes its version banner through C stdio, which is fully buffered on a pipe
    # and would otherwise be flushed when a rank exits, after the line.  Every rank flushes, all meet, then rank 0 prints.
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if dist.is_initialized():
        dist.barrier()
    if out is not None:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
