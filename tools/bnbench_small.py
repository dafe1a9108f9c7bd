#!/usr/bin/env python3
"""coin_bn_apply_fwd / coin_bn_bwd at the backbone's map sizes (launch-shape tuning of the row-walk kernels, round 6)."""
import json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from coin_amd import kernels as K
from bnbench import timeit

res = {}
for name, shape, pool, resid in [("l3 256@50x83", (4, 50, 83, 256), 1, False), ("l3 1024@50x83 res", (4, 50, 83, 1024), 1, True), ("l2 128@100x167", (4, 100, 167, 128), 1, False),
                                 ("l2 512@100x167 res", (4, 100, 167, 512), 1, True), ("l2.0 128@200x333 pool", (4, 200, 333, 128), 2, False), ("res5 512@14x14", (2048, 14, 14, 512), 1, False)]:
    n, h, w, c = shape
    x = torch.randn(shape, device="cuda").to(torch.bfloat16)
    gam, bet = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda")
    r = torch.randn(shape, device="cuda").to(torch.bfloat16) if resid else None
    mean, rstd = K.bn_stats(x, 1e-5, 0.1)
    if resid:
        t_apply = timeit(lambda: K.bn_apply_fwd(x, mean, rstd, gam, bet, r, True, pool, want_mask=True), iters=50)
        y, mk = K.bn_apply_fwd(x, mean, rstd, gam, bet, r, True, pool, want_mask=True)
        dy = torch.randn_like(y)
        t_bwd = timeit(lambda: K.bn_bwd(x, dy, None, mean, rstd, gam, bet, True, pool, True, mask=mk), iters=50)
    else:
        t_apply = timeit(lambda: K.bn_apply_fwd(x, mean, rstd, gam, bet, None, True, pool), iters=50)
        y = K.bn_apply_fwd(x, mean, rstd, gam, bet, None, True, pool)
        dy = torch.randn_like(y)
        t_bwd = timeit(lambda: K.bn_bwd(x, dy, None, mean, rstd, gam, bet, True, pool, False), iters=50)
    res[name] = {"apply_us": round(t_apply * 1e3, 1), "bwd_us": round(t_bwd * 1e3, 1), "x_MB": round(x.numel() * 2 / 1e6, 1)}
print(json.dumps(res))
