#!/usr/bin/env python3
"""cProfile of one targetDET step (development tool)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from coin_amd.config import get_cfg
from coin_amd.data.synthetic import synthetic_offline_detections
from coin_amd.engine import CoinTrainer

step_two = "--step-two" in sys.argv
cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs", "coin", "GDINO", "foggy_synthetic.yaml"))
cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", 3, "AMD.SYNTHETIC.NUM_IMAGES", 3, "AMD.TEXT_TEMPLATES", 4, "MODEL.DEVICE", "cuda:0",
                     "CLOUD.BURN_UP_STEP", 0 if step_two else 10 ** 9, "CLOUD.PROTOTYPE_UPDATE_START", 0])
torch.manual_seed(cfg.SEED)
tr = CoinTrainer(cfg)
real_forward, g = tr.offline_teacher.forward, torch.Generator().manual_seed(7)


def teacher(batched_inputs, branch=None, **kw):
    real_forward(batched_inputs, branch=branch, **kw)
    return [synthetic_offline_detections(tr.model_CLOUD.entry(d["file_name"]), g, device="cuda:0") for d in batched_inputs]


tr.offline_teacher.forward = teacher
tr.max_iter = 10 ** 9
for _ in range(3):
    tr.run_step()
    tr.prepare_next()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.run_step()
    tr.prepare_next()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(70)
st.sort_stats("tottime").print_stats(40)
