#!/usr/bin/env python3
"""Development probe for coin_amd/graphs.py (GPU box): which stretch captures / replays, with a Python traceback on a crash.

    python -X faulthandler tools/graph_probe.py blocks|trunk|backbone|both [steps]
"""
import faulthandler
import os
import sys
import time

faulthandler.enable()
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8


def say(*a):
    print(*a, flush=True)


from coin_amd import graphs as G
from coin_amd import layers as L

_cap = G.GraphedSegment._capture


def traced_capture(self, inputs):
    say(f"[{self.name}] capture begin", [tuple(x.shape) for x in inputs])
    t0 = time.perf_counter()
    ent = _cap(self, inputs)
    say(f"[{self.name}] capture done in {time.perf_counter() - t0:.2f} s, bwd={'yes' if ent.bwd is not None else 'no'}")
    return ent


G.GraphedSegment._capture = traced_capture

if os.environ.get("PROBE_BWD_MODE"):
    G.CAPTURE_MODE["bwd"] = os.environ["PROBE_BWD_MODE"]
if os.environ.get("PROBE_FWD_MODE"):
    G.CAPTURE_MODE["fwd"] = os.environ["PROBE_FWD_MODE"]
say("capture modes", G.CAPTURE_MODE)

if mode in ("linear", "convgemm", "bn"):
    # bisect: which kind of node breaks the backward capture
    L.CONV_GEMM.update(enabled=True, min_rows=0, wgrad=True)
    if mode == "linear":
        lin = torch.nn.Linear(1024, 1024).cuda()
        fn, params = (lambda x: torch.nn.functional.linear(x, lin.weight, lin.bias)), (lambda: list(lin.parameters()))
        mk = lambda: torch.randn(512, 1024, device="cuda").requires_grad_(True)
    elif mode == "convgemm":
        conv = torch.nn.Conv2d(1024, 256, 1, bias=False).cuda().to(memory_format=torch.channels_last)
        fn, params = (lambda x: L.conv2d_gemm(x, conv)[0]), (lambda: list(conv.parameters()))
        mk = lambda: torch.randn(16, 1024, 14, 14, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    else:
        bn = torch.nn.BatchNorm2d(1024).cuda().train()
        fn, params = (lambda x: L.bn_act(x, bn, True)), (lambda: list(bn.parameters()))
        mk = lambda: torch.randn(16, 1024, 14, 14, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    seg = G.GraphedSegment(mode, fn, params)
    for i in range(steps):
        x = mk()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode != "linear"):
            y = seg(x)
        y.float().sum().backward()
        torch.cuda.synchronize()
        say("step", i, "ok", float(x.grad.float().abs().sum()), G.STATS)
    say("RESULT", mode, "survived")
    sys.exit(0)

if mode == "blocks":
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
    import seeded
    from coin_amd.modeling.backbone import Bottleneck

    L.CONV_GEMM.update(enabled=True, min_rows=0, wgrad=True)
    b = torch.nn.Sequential(seeded.fill_module(Bottleneck(1024, 256, 2), 5), seeded.fill_module(Bottleneck(1024, 256, 1), 6)).cuda().to(memory_format=torch.channels_last).train()
    seg = G.GraphedSegment("blocks", lambda x: b(x), lambda: list(b.parameters()))
    for i in range(steps):
        x = torch.randn(16, 1024, 14, 14, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = seg(x)
        say("step", i, "fwd ok")
        y.float().sum().backward()
        torch.cuda.synchronize()
        say("step", i, "bwd ok", float(x.grad.float().abs().sum()), G.STATS)
    sys.exit(0)

from bench import build_cfg
from coin_amd.engine import PRETrainer

torch.backends.cudnn.benchmark = True
cfg = build_cfg(1, "cuda:0", "bf16")
torch.manual_seed(cfg.SEED)
tr = PRETrainer(cfg)
tr.model.step_graphs = mode in ("backbone", "both")
tr.model.roi_heads.step_graphs = mode in ("trunk", "both")
say("mode", mode, "backbone graphs", tr.model.step_graphs, "trunk graphs", tr.model.roi_heads.step_graphs)
for i in range(steps):
    t0 = time.perf_counter()
    rec = tr.run_step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    say("step", i, f"host {th * 1e3:.1f} ms total {(time.perf_counter() - t0) * 1e3:.1f} ms loss {float(sum(rec.values())):.4f}", G.STATS)
torch.cuda.synchronize()
t0 = time.perf_counter()
hs = []
for i in range(20):
    th = time.perf_counter()
    tr.run_step()
    hs.append((time.perf_counter() - th) * 1e3)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
say(f"RESULT mode={mode} ms_per_step={dt * 1e3:.2f} views_per_s={4 / dt:.1f} host_ms_median={sorted(hs)[10]:.1f}", G.STATS)
# pure host enqueue time: the device is idle when a step starts, nothing throttles the host
hs, l0 = [], G.STATS["launch_ms"]
for i in range(10):
    torch.cuda.synchronize()
    th = time.perf_counter()
    tr.run_step()
    hs.append((time.perf_counter() - th) * 1e3)
torch.cuda.synchronize()
say(f"HOST mode={mode} enqueue_ms_median={sorted(hs)[5]:.1f} min={min(hs):.1f} graph_launch_ms_per_step={(G.STATS['launch_ms'] - l0) / 10:.2f}")
