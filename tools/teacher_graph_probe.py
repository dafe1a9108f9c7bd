#!/usr/bin/env python3
"""Does the teacher's graph replay the same bits whatever the allocator did in between?  (development probe)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from coin_amd.config import get_cfg
from coin_amd.engine import PRETrainer

torch.backends.cudnn.benchmark = True
cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs", "coin", "PRETRAINS", "CLIPDET_synthetic.yaml"))
cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", 1, "AMD.SYNTHETIC.NUM_IMAGES", 1, "AMD.COMPUTE_DTYPE", "bf16", "AMD.TEXT_TEMPLATES", 2,
                     "MODEL.DEVICE", "cuda:0", "AMD.STEP_GRAPHS", False, "AMD.SYNTHETIC.HEIGHT", 608, "AMD.SYNTHETIC.WIDTH", 800])
torch.manual_seed(7)
tr = PRETrainer(cfg)
model = tr.model.eval()
strong, weak = next(tr._data_loader_iter)
batch = [dict(d) for d in weak]


def run(graph):
    with torch.no_grad():
        assert model.inference_begin(batch, branch="test", graph=graph)
        _, boxes, probs, _ = model._begun
        model._begun = None
        torch.cuda.synchronize()
        return boxes.clone(), probs.clone()


def cmp(tag, a, b):
    db, dp = (a[0] - b[0]).abs(), (a[1] - b[1]).abs()
    rows = int((db.amax(dim=-1) > 0).sum())
    print(f"TEACHER-GRAPH {tag:44s}: boxes max diff {float(db.max()):.3e} ({rows} of {db.shape[0] * db.shape[1]} rows differ), probs max diff {float(dp.max()):.3e}", flush=True)


e0 = run(False); e1 = run(False)
cmp("eager vs eager", e0, e1)
for _ in range(3):
    g0 = run(True)
assert model._graphs and not model.graph_failed
g1 = run(True)
cmp("replay vs replay", g0, g1)
cmp("replay vs eager", g1, e0)
junk = [torch.full((1 << 26,), float("nan"), device="cuda") for _ in range(8)]
g2 = run(True)
cmp("replay after 2 GiB of NaN allocations", g2, g1)
del junk
torch.cuda.synchronize(); torch.cuda.empty_cache()
g3 = run(True)
cmp("replay after empty_cache", g3, g1)
junk = [torch.full((1 << 26,), float("nan"), device="cuda") for _ in range(16)]
g4 = run(True)
cmp("replay after empty_cache + 4 GiB of NaN", g4, g1)
e2 = run(False)
cmp("eager afterwards vs eager before", e2, e0)
cmp("last replay vs eager", g4, e0)
