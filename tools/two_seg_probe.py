#!/usr/bin/env python3
"""Two graphed stretches in series (development probe): is a stretch's pending backward disturbed when the NEXT stretch is captured between its
forward replay and its backward replay?   python tools/two_seg_probe.py [same|stagger]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import seeded
from coin_amd import graphs as G
from coin_amd import layers as L
from coin_amd.modeling.backbone import Bottleneck
from coin_amd.solver.build import FusedSGD
mode = sys.argv[1] if len(sys.argv) > 1 else "same"
L.CONV_GEMM.update(enabled=True, min_rows=0, wgrad=True)
def mk(seed):
    a = torch.nn.Sequential(seeded.fill_module(Bottleneck(1024, 256, 2), seed), seeded.fill_module(Bottleneck(1024, 256, 1), seed + 1)).cuda().to(memory_format=torch.channels_last).train()
    b = torch.nn.Sequential(seeded.fill_module(Bottleneck(1024, 256, 1), seed + 2), seeded.fill_module(Bottleneck(1024, 256, 1), seed + 3)).cuda().to(memory_format=torch.channels_last).train()
    return a, b
ea, eb = mk(5)
ga, gb = mk(5)
oe = FusedSGD([{"params": [p]} for m in (ea, eb) for p in m.parameters()], lr=1e-3, momentum=0.9, weight_decay=1e-4)
og = FusedSGD([{"params": [p]} for m in (ga, gb) for p in m.parameters()], lr=1e-3, momentum=0.9, weight_decay=1e-4)
sa = G.GraphedSegment("A", lambda x: ga(x), lambda: list(ga.parameters()), lambda: list(ga.buffers()))
sb = G.GraphedSegment("B", lambda x: gb(x), lambda: list(gb.parameters()), lambda: list(gb.buffers()))
gen = torch.Generator(device="cuda").manual_seed(3)
for step in range(7):
    x0 = torch.randn(16, 1024, 14, 14, device="cuda", generator=gen).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(16, 1024, 7, 7, device="cuda", generator=gen).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = []
    for tag, fa, fb, mods, opt in (("eager", lambda x: ea(x), lambda x: eb(x), (ea, eb), oe), ("graph", sa, sb, (ga, gb), og)):
        x = x0.clone().requires_grad_(True)
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if tag == "graph" and mode == "stagger" and step == 2:
                h = fa(x); y = gb(h)          # B stays eager this step: its capture comes one step later
            else:
                y = fb(fa(x))
        y.backward(gy)
        res.append((y.detach().clone(), x.grad.clone(), [p.grad.clone() for m in mods for p in m.parameters()]))
        opt.step()
        if tag == "graph":
            G.step_done()
    (ye, gxe, gpe), (yg, gxg, gpg) = res
    bad = [i for i, (u, v) in enumerate(zip(gpe, gpg)) if not torch.equal(u, v)]
    print(f"step {step}: y equal {torch.equal(ye, yg)}, gx equal {torch.equal(gxe, gxg)}, param grads differing: {len(bad)} of {len(gpe)} (first {bad[:4]})", {k: G.STATS[k] for k in ("captures", "replays")}, flush=True)
