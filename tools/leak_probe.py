#!/usr/bin/env python3
"""Which part of the backbone's graphs depends on memory that is not theirs?  (development probe)
A stretch is captured, the allocator's cache is emptied and refilled with +inf, the stretch is replayed: a correct graph still reproduces its
eager twin; one that reads a stale address does not."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import seeded
from coin_amd import graphs as G
from coin_amd import layers as L
from coin_amd.modeling.backbone import ModifiedResNet
torch.backends.cudnn.benchmark = True
L.CONV_GEMM.update(enabled=True, wgrad=True)

def case(name, pick, in_ch, hw, min_rows):
    L.CONV_GEMM["min_rows"] = min_rows
    net = ModifiedResNet((3, 4, 6, 3), 64, ("res4",), 2)
    seeded.fill_module(net, 11)
    net = net.cuda().to(memory_format=torch.channels_last).train()
    mod = pick(net)
    params = [p for p in mod.parameters() if p.requires_grad]
    seg = G.GraphedSegment(name, lambda x: mod(x), lambda: params, lambda: list(mod.buffers()))
    gen = torch.Generator(device="cuda").manual_seed(3)
    x0 = torch.randn(4, in_ch, hw[0], hw[1], device="cuda", generator=gen).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    def run(fn, graphs):
        for p in params: p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            x = x0.clone().requires_grad_(True)
            y = fn(x)
        y.backward(torch.ones_like(y) * 0.01)
        return x.grad.float().clone(), [p.grad.float().clone() for p in params]
    for i in range(4):           # eager, eager(announce), capture, replay
        run(seg, True); G.step_done()
    assert len(seg.graphs) == 1, "not captured"
    ref = run(lambda x: mod(x), False)
    good = run(seg, True); G.step_done()
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    junk = [torch.full((1 << 27,), float("inf"), device="cuda") for _ in range(16)]   # 8 GiB of +inf over the released ranges
    torch.cuda.synchronize()
    bad = run(seg, True); G.step_done()
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))
    e_good = max([rel(good[0], ref[0])] + [rel(a, b) for a, b in zip(good[1], ref[1])])
    e_bad = max([rel(bad[0], ref[0])] + [rel(a, b) if torch.isfinite(a).all() else float("inf") for a, b in zip(bad[1], ref[1])])
    print(f"LEAK {name:28s} min_rows={min_rows:6d}: replay vs eager before empty_cache {e_good:.2e}, after {e_bad:.2e}", flush=True)
    del junk

for mr in (32768, 0):
    case("layer2", lambda n: n.layer2, 256, (200, 333), mr)
    case("layer3", lambda n: n.layer3, 512, (100, 167), mr)
    case("layer3[0]", lambda n: n.layer3[0], 512, (100, 167), mr)
    case("layer3[1]", lambda n: n.layer3[1], 1024, (50, 83), mr)
