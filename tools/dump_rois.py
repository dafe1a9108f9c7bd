#!/usr/bin/env python3
"""The RoIs of a real bench step (RPN proposals ++ teacher boxes, as sampled): run the pre-train bench model for a few steps, capture the
[R, 5] tensor RoIAlign forward is called with, save it as a small fixture for tools/roibench.py / kbench.py (`--rois FILE`) and print its
box-size statistics.  The uniform 32-400 px boxes those tools draw by default are NOT representative (profiles/README.md, round 4)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from coin_amd import kernels as K
from coin_amd import layers as L
from coin_amd.engine import PRETrainer

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "real_rois.pt")
torch.backends.cudnn.benchmark = True
cfg = bench.build_cfg(1, "cuda:0", "bf16")
torch.manual_seed(cfg.SEED)
tr = PRETrainer(cfg)
seen = []
real = K.roi_align_fwd


def spy(feat, rois, *a, **k):
    seen.append(rois.detach().float().cpu().clone())
    return real(feat, rois, *a, **k)


K.roi_align_fwd = spy
for _ in range(6):
    tr.run_step()
torch.cuda.synchronize()
K.roi_align_fwd = real
r = seen[-1]
os.makedirs(os.path.dirname(out), exist_ok=True)
torch.save(r, out)
w, h = r[:, 3] - r[:, 1], r[:, 4] - r[:, 2]
q = lambda t: [round(float(v), 1) for v in torch.quantile(t, torch.tensor([0.05, 0.25, 0.5, 0.75, 0.95]))]
print("rois", tuple(r.shape), "per image", torch.bincount(r[:, 0].long()).tolist())
print("width  px quantiles 5/25/50/75/95 %:", q(w))
print("height px quantiles 5/25/50/75/95 %:", q(h))
print("sampling grid (ceil(h/16/14) x ceil(w/16/14)) histogram:", torch.bincount((torch.ceil(h / 224).clamp(min=1) * 10 + torch.ceil(w / 224).clamp(min=1)).long()).nonzero().flatten().tolist())
print("saved", out)
