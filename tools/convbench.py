#!/usr/bin/env python3
"""Micro-benchmark (development tool): MIOpen 1x1 / 3x3 convolutions on channels-last bf16 vs the same contraction as a
library GEMM (hipBLASLt through torch.mm), forward / dgrad / wgrad, at the res5 and backbone shapes of the benchmark step."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


res = {}
shapes = [  # name, N, H, W, Cin, Cout
    ("l4.0.conv1", 2048, 14, 14, 1024, 512), ("l4.0.conv3", 2048, 7, 7, 512, 2048), ("l4.0.down", 2048, 7, 7, 1024, 2048),
    ("l4.1.conv1", 2048, 7, 7, 2048, 512), ("l3.conv1", 4, 50, 83, 1024, 256), ("l3.conv3", 4, 50, 83, 256, 1024),
    ("l2.conv1", 4, 100, 166, 512, 128), ("l2.conv3", 4, 100, 166, 128, 512),
]
for name, n, h, w, ci, co in shapes:
    x = torch.randn(n, ci, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 1, 1, device="cuda", dtype=torch.bfloat16) * 0.02).contiguous(memory_format=torch.channels_last)
    m = n * h * w
    flop = 2.0 * m * ci * co
    xr, wr = x.detach().requires_grad_(True), wt.detach().requires_grad_(True)
    y = F.conv2d(xr, wr)
    gy = torch.randn_like(y)
    t_f = timeit(lambda: F.conv2d(x, wt))
    t_b = timeit(lambda: torch.autograd.grad(F.conv2d(xr, wr), (xr, wr), gy)) - t_f
    x2 = x.permute(0, 2, 3, 1).reshape(m, ci)
    w2 = wt.reshape(co, ci)
    g2 = gy.permute(0, 2, 3, 1).reshape(m, co)
    assert x2.is_contiguous() and g2.is_contiguous()
    t_mf = timeit(lambda: x2 @ w2.t())
    t_md = timeit(lambda: g2 @ w2)
    t_mw = timeit(lambda: g2.t() @ x2)
    ref = F.conv2d(x, wt).permute(0, 2, 3, 1).reshape(m, co).float()
    err = float(((x2 @ w2.t()).float() - ref).abs().max() / ref.abs().max())
    res[name] = {"M": m, "Cin": ci, "Cout": co, "conv_fwd_ms": t_f, "conv_fwd_TF": flop / t_f / 1e9, "conv_bwd_ms": t_b, "conv_bwd_TF": 2 * flop / t_b / 1e9,
                 "mm_fwd_ms": t_mf, "mm_fwd_TF": flop / t_mf / 1e9, "mm_dgrad_ms": t_md, "mm_dgrad_TF": flop / t_md / 1e9,
                 "mm_wgrad_ms": t_mw, "mm_wgrad_TF": flop / t_mw / 1e9, "rel_err": err}
    print(name, json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res[name].items()}), flush=True)

# 3x3 at res5 for reference
for name, n, h, w, c in [("l4.0.conv2", 2048, 14, 14, 512), ("l4.1.conv2", 2048, 7, 7, 512), ("rpn.conv", 4, 50, 83, 1024)]:
    x = torch.randn(n, c, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(c, c, 3, 3, device="cuda", dtype=torch.bfloat16) * 0.02).contiguous(memory_format=torch.channels_last)
    flop = 2.0 * n * h * w * c * c * 9
    xr, wr = x.detach().requires_grad_(True), wt.detach().requires_grad_(True)
    gy = torch.randn_like(F.conv2d(xr, wr, padding=1))
    t_f = timeit(lambda: F.conv2d(x, wt, padding=1))
    t_b = timeit(lambda: torch.autograd.grad(F.conv2d(xr, wr, padding=1), (xr, wr), gy)) - t_f
    print(name, json.dumps({"conv_fwd_ms": round(t_f, 3), "conv_fwd_TF": round(flop / t_f / 1e9, 1), "conv_bwd_ms": round(t_b, 3),
                            "conv_bwd_TF": round(2 * flop / t_b / 1e9, 1)}), flush=True)
