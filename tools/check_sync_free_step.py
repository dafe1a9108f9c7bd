#!/usr/bin/env python3
"""GPU smoke of the sync-free step_two path (cfg.AMD.SYNC_FREE_STEP): 3 CoinTrainer steps at full size, losses must be finite."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from coin_amd.config import get_cfg
from coin_amd.data.synthetic import synthetic_offline_detections
from coin_amd.engine import CoinTrainer

cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs", "coin", "GDINO", "foggy_synthetic.yaml"))
cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", 2, "AMD.SYNTHETIC.NUM_IMAGES", 2, "AMD.TEXT_TEMPLATES", 2, "MODEL.DEVICE", "cuda:0",
                     "CLOUD.BURN_UP_STEP", 0, "CLOUD.PROTOTYPE_UPDATE_START", 0, "CLOUD.CLS_B_THRESH", 0.2, "AMD.SYNC_FREE_STEP", True])
torch.manual_seed(11)
tr = CoinTrainer(cfg)
real_forward, g = tr.offline_teacher.forward, torch.Generator().manual_seed(7)


def teacher(batched_inputs, branch=None, **kw):
    real_forward(batched_inputs, branch=branch, **kw)
    return [synthetic_offline_detections(tr.model_CLOUD.entry(d["file_name"]), g, device="cuda:0") for d in batched_inputs]


tr.offline_teacher.forward = teacher
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec = tr.run_step()
    torch.cuda.synchronize()
    vals = {k: round(float(v), 4) for k, v in rec.items()}
    assert all(v == v and abs(v) < 1e6 for v in vals.values()), vals
    print(f"step {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms", vals if i == 3 else "")
print("sync-free step path ok")
