#!/usr/bin/env python3
"""Development tool: run the same pre-train forward twice from the same seed in one process and report the first stage whose
tensors differ (proposals -> sampled RoIs -> losses).  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from coin_amd.config import get_cfg
from coin_amd.engine import PRETrainer
from coin_amd import kernels as K, box_ops

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs", "coin", "PRETRAINS", "CLIPDET_synthetic.yaml"))
cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", 1, "AMD.SYNTHETIC.NUM_IMAGES", 1, "AMD.SYNTHETIC.HEIGHT", 384, "AMD.SYNTHETIC.WIDTH", 640,
                     "AMD.TEXT_TEMPLATES", 2, "MODEL.DEVICE", "cuda:0", "AMD.COMPUTE_DTYPE", dtype])
torch.manual_seed(5)
tr = PRETrainer(cfg)
strong, weak = next(tr._data_loader_iter)
strong, weak = tr.set_boxes([strong, weak])
batch = strong + weak
rec = {}
real_sample = K.sample_labels
real_match = K.anchor_match
rh = tr.model.roi_heads
real_packed = rh.sample_packed


def spy_sample(cls, keys, *a):
    out = real_sample(cls, keys, *a)
    rec.setdefault("sample", []).append((cls.clone(), keys.clone(), out.clone()))
    again = real_sample(cls, keys, *a)
    if not torch.equal(out, again):
        print("  !! sample_labels not reproducible on identical inputs:", tuple(cls.shape), cls.dtype, int((out != again).sum()), "elements differ")
    return out


def spy_match(*a, **k):
    out = real_match(*a, **k)
    rec.setdefault("match", []).append(tuple(None if t is None else t.clone() for t in out))
    return out


def spy_packed(proposals, targets):
    ps = real_packed(proposals, targets)
    rec.setdefault("packed", []).append((proposals.boxes.clone(), proposals.valid.clone(), ps.boxes.clone(), ps.gt_classes.clone()))
    return ps


K.sample_labels = spy_sample
K.anchor_match = spy_match
rh.sample_packed = spy_packed
runs = []
for r in range(2):
    rec = {}
    torch.manual_seed(6)
    with torch.no_grad():
        losses = tr.model(batch, branch="pre_train", update_prototype=False)
    torch.cuda.synchronize()
    runs.append((rec, {k: float(v) for k, v in losses.items()}))
a, b = runs
print("losses run 0:", a[1])
print("losses run 1:", b[1])
for key in ("match", "sample", "packed"):
    for i, (ta, tb) in enumerate(zip(a[0].get(key, []), b[0].get(key, []))):
        for j, (x, y) in enumerate(zip(ta, tb)):
            if x is None:
                continue
            same = torch.equal(x, y)
            print(f"{key}[{i}][{j}] shape {tuple(x.shape)} {x.dtype}: {'same' if same else 'DIFFERENT (%d elements)' % int((x != y).sum())}")
