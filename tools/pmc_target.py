#!/usr/bin/env python3
"""Target for the rocprofv3 --pmc passes: a few launches of the hand-written HBM-bound kernels at BASELINE sizes."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from coin_amd import kernels as K
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import rois_like_bench

g = torch.Generator().manual_seed(0)
n, c, h, w, r = 4, 1024, 50, 83, 2048
rois = rois_like_bench(n, 512, g).cuda()
feat = torch.randn(n, h, w, c, device="cuda").to(torch.bfloat16)
go = torch.randn(r, 14, 14, c, device="cuda").to(torch.bfloat16)
x = torch.randn(2048, 7, 7, 2048, device="cuda").to(torch.bfloat16)
res = torch.randn_like(x)
dy = torch.randn_like(x)
gam, bet = torch.rand(2048, device="cuda") + 0.5, torch.randn(2048, device="cuda")
for _ in range(3):
    out = K.roi_align_fwd(feat, rois, (14, 14), 1 / 16.0)
    gf = K.roi_align_bwd(go, rois, (n, h, w, c), 1 / 16.0)
    mean, rstd = K.bn_stats(x, 1e-5, 0.1)
    y = K.bn_apply_fwd(x, mean, rstd, gam, bet, res, True, 1)
    K.bn_bwd(x, dy, y, mean, rstd, gam, bet, True, 1, True)
torch.cuda.synchronize()
