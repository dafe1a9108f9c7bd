#!/usr/bin/env python3
"""Micro-benchmark: fused BN kernels vs torch (MIOpen) at res5 shapes (4 views x 512 RoIs)."""
import os, sys, json
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import torch.nn.functional as F
from coin_amd import kernels as K, layers as L

def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

res = {}
for name, shape, pool, resid in [("c512_14", (2048, 14, 14, 512), 1, False), ("c512_14_pool", (2048, 14, 14, 512), 2, False),
                                 ("c2048_7_res", (2048, 7, 7, 2048), 1, True), ("c512_7", (2048, 7, 7, 512), 1, False)]:
    n, h, w, c = shape
    x = torch.randn(shape, device="cuda").to(torch.bfloat16)
    gam, bet = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda")
    r = torch.randn(shape, device="cuda").to(torch.bfloat16) if resid else None
    gb = x.numel() * 2 / 1e9
    t_stats = timeit(lambda: K.bn_stats(x, 1e-5, 0.1))
    mean, rstd = K.bn_stats(x, 1e-5, 0.1)
    t_apply = timeit(lambda: K.bn_apply_fwd(x, mean, rstd, gam, bet, r, True, pool))
    y = K.bn_apply_fwd(x, mean, rstd, gam, bet, r, True, pool)
    dy = torch.randn_like(y)
    t_bwd = timeit(lambda: K.bn_bwd(x, dy, y if pool == 1 else None, mean, rstd, gam, bet, True, pool, resid))
    t_apply_m = t_bwd_m = None
    if resid:  # the ReLU bit mask instead of the saved output
        t_apply_m = timeit(lambda: K.bn_apply_fwd(x, mean, rstd, gam, bet, r, True, pool, want_mask=True))
        _, mk = K.bn_apply_fwd(x, mean, rstd, gam, bet, r, True, pool, want_mask=True)
        t_bwd_m = timeit(lambda: K.bn_bwd(x, dy, None, mean, rstd, gam, bet, True, pool, resid, mask=mk))
    # torch reference
    xt = x.permute(0, 3, 1, 2).detach().requires_grad_(True)
    bn = torch.nn.BatchNorm2d(c).cuda()
    def tf():
        o = F.relu(bn(xt) + (r.permute(0, 3, 1, 2) if resid else 0))
        return F.avg_pool2d(o, 2) if pool == 2 else o
    t_tf = timeit(tf)
    o = tf(); go = torch.randn_like(o)
    def tb():
        o = tf(); o.backward(go)
    t_tfb = timeit(tb)
    res[name] = {"x_GB": gb, "stats_ms": t_stats, "stats_GBps": gb / t_stats * 1e3, "apply_ms": t_apply, "bwd_ms": t_bwd, "apply_mask_ms": t_apply_m, "bwd_mask_ms": t_bwd_m,
                 "ours_fwd_ms": t_stats + t_apply, "ours_fwdbwd_ms": t_stats + t_apply + t_bwd, "torch_fwd_ms": t_tf, "torch_fwdbwd_ms": t_tfb}
print(json.dumps(res, indent=1))
