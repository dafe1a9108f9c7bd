#!/usr/bin/env python3
"""targetDET step: host enqueue time per phase with the device idle at the start of each step (development probe).
    python tools/prof_td.py [--step-two] [--images 3]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
ap = argparse.ArgumentParser(); ap.add_argument("--step-two", action="store_true"); ap.add_argument("--images", type=int, default=3)
args = ap.parse_args()
import torch
from bench_targetdet import build_trainer
from coin_amd import graphs as G
tr = build_trainer("foggy", args.images, args.step_two)
for _ in range(10):
    tr.run_step(); tr.prepare_next()
torch.cuda.synchronize()
rows = []
for i in range(12):
    torch.cuda.synchronize()
    l0 = G.STATS["launch_ms"]
    t0 = time.perf_counter(); tr.run_step(); t1 = time.perf_counter(); tr.prepare_next(); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t0) * 1e3, G.STATS["launch_ms"] - l0))
med = lambda k: sorted(r[k] for r in rows)[len(rows) // 2]
print(f"TD step_two={args.step_two} graphs={G.ENABLED['on']}: run_step host {med(0):.1f} ms, prepare_next host {med(1):.1f} ms, step complete (no pipelining) {med(2):.1f} ms, graph launches {med(3):.2f} ms", G.STATS, flush=True)
# pipelined, every step timed, with the allocator's device-level activity (a hipMalloc / hipFree in the loop synchronises the device)
ms = lambda: torch.cuda.memory_stats()
per, t0 = [], time.perf_counter()
for _ in range(24):
    a0 = (ms().get("num_device_alloc", 0), ms().get("num_device_free", 0), ms().get("num_alloc_retries", 0))
    ts = time.perf_counter(); tr.run_step(); tm = time.perf_counter(); tr.prepare_next(); te = time.perf_counter()
    a1 = (ms().get("num_device_alloc", 0), ms().get("num_device_free", 0), ms().get("num_alloc_retries", 0))
    per.append((round((tm - ts) * 1e3, 1), round((te - tm) * 1e3, 1), tuple(b - a for a, b in zip(a0, a1))))
torch.cuda.synchronize()
print(f"TD pipelined {(time.perf_counter() - t0) / 24 * 1e3:.1f} ms/step; per step (run_step ms, prepare_next ms, (device allocs, frees, retries)):", per, flush=True)
print("TD reserved GB", round(torch.cuda.memory_reserved() / 2**30, 1), "allocated GB", round(torch.cuda.memory_allocated() / 2**30, 1), G.STATS, flush=True)
