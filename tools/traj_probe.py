#!/usr/bin/env python3
"""Loss trajectory of the small pre-train configuration, one process per mode (development probe):
    COIN_STEP_GRAPHS=0|1 python tools/traj_probe.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from coin_amd import graphs as G
from coin_amd.config import get_cfg
from coin_amd.engine import PRETrainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs", "coin", "PRETRAINS", "CLIPDET_synthetic.yaml"))
cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", 1, "AMD.SYNTHETIC.NUM_IMAGES", 1, "AMD.COMPUTE_DTYPE", "bf16", "AMD.TEXT_TEMPLATES", 2,
                     "MODEL.DEVICE", "cuda:0", "AMD.SYNTHETIC.HEIGHT", 608, "AMD.SYNTHETIC.WIDTH", 800])
torch.manual_seed(21)
tr = PRETrainer(cfg)
with torch.no_grad():
    for n, p in tr.model.named_parameters():
        if n.endswith("bn3.weight"):
            p.fill_(0.5)
for i in range(steps):
    torch.manual_seed(1000 + i)
    rec = tr.run_step()
    w = tr.model.backbone.encoder.visual.layer3[0].conv2.weight
    print("STEP", i, " ".join(f"{k}={float(v):.6f}" for k, v in rec.items()), f"w_l3={float(w.detach().float().norm()):.7f} w_head={float(tr.model.roi_heads.box_predictor.trans[0].weight.detach().norm()):.7f}", G.STATS["replays"], flush=True)
