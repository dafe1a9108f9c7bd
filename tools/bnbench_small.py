#!/usr/bin/env python3
"""coin_bn_apply_fwd / coin_bn_bwd at the backbone's map sizes (launch-shape tuning of the row-walk kernels, round 6)."""
import json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from coin_amd import _lib
if os.environ.get("BN_LAB"):   # the lab library: coin_lab_set_bn_grid (workgroups of the row-walk kernels); before anything loads the product one
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab", "libcoin_hip_lab.so")
from coin_amd import kernels as K


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

caps = [int(c) for c in os.environ.get("BN_CAPS", "0").split(",")]
if os.environ.get("BN_PARTS"):   # experiment: more partial sums in the backward's reduction pass (the workspace is sized for them here)
    K.BN_MAX_PARTS = int(os.environ["BN_PARTS"])
    _lib.lib().coin_lab_set_bn_parts(int(os.environ["BN_PARTS"]))
res = {}
for cap in caps:
  if cap:
    _lib.lib().coin_lab_set_bn_grid(cap)
  for name, shape, pool, resid in [("l3 256@50x83", (4, 50, 83, 256), 1, False), ("l3 1024@50x83 res", (4, 50, 83, 1024), 1, True), ("l2 128@100x167", (4, 100, 167, 128), 1, False),
                                   ("l2 512@100x167 res", (4, 100, 167, 512), 1, True), ("l2.0 128@200x333 pool", (4, 200, 333, 128), 2, False), ("res5 512@14x14", (2048, 14, 14, 512), 1, False), ("res5 2048@7x7 res", (2048, 7, 7, 2048), 1, True), ("res5 512@14x14 pool", (2048, 14, 14, 512), 2, False)]:
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
      res[f"{name} cap={cap}"] = {"apply_us": round(t_apply * 1e3, 1), "bwd_us": round(t_bwd * 1e3, 1), "x_MB": round(x.numel() * 2 / 1e6, 1)}
print(json.dumps(res))
