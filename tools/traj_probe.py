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
if os.environ.get("BENCHCFG") == "1":   # exactly the benchmark's trainer and seeding (bench.py)
    import bench
    torch.backends.cudnn.benchmark = True
    cfg = bench.build_cfg(1, "cuda:0", "bf16")
    torch.manual_seed(cfg.SEED)
    tr = PRETrainer(cfg)
    reseed = False
else:
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "coin", "PRETRAINS", "CLIPDET_synthetic.yaml"))
    cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", 1, "AMD.SYNTHETIC.NUM_IMAGES", 1, "AMD.COMPUTE_DTYPE", "bf16", "AMD.TEXT_TEMPLATES", 2,
                         "MODEL.DEVICE", "cuda:0", "AMD.SYNTHETIC.HEIGHT", 608, "AMD.SYNTHETIC.WIDTH", 800])
    torch.manual_seed(21)
    tr = PRETrainer(cfg)
    reseed = True
    with torch.no_grad():
        for n, p in tr.model.named_parameters():
            if n.endswith("bn3.weight"):
                p.fill_(0.5)
seg = os.environ.get("COIN_SEG", "both")
tr.model.step_graphs = G.ENABLED["on"] and seg in ("both", "backbone")
tr.model.roi_heads.step_graphs = G.ENABLED["on"] and seg in ("both", "trunk")
if os.environ.get("NO_LOOKAHEAD") == "1":
    tr.model.overlap_streams = False
groups = {}
for n, p in tr.model.named_parameters():
    if p.requires_grad:
        key = ".".join(n.split(".")[:5]) if n.startswith("backbone") else ".".join(n.split(".")[:3])
        groups.setdefault(key, []).append(p)
junk = []
for i in range(steps):
    if os.environ.get("PROBE_EMPTY_CACHE") and i == int(os.environ["PROBE_EMPTY_CACHE"]):
        # after the backbone's capture: release the allocator's cache and let fresh allocations take the freed address ranges
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        junk = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(12)]   # 12 GiB of NaNs
        torch.cuda.synchronize()
        print("PROBE emptied the cache and allocated NaN blocks", flush=True)
    if reseed:
        torch.manual_seed(1000 + i)
    rec = tr.run_step()
    if i in (1, 2, 3):
        print("PARAMS", i, " ".join(f"{k}={sum(float(p.detach().double().abs().sum()) for p in ps):.8e}" for k, ps in sorted(groups.items())), flush=True)
    w = tr.model.backbone.encoder.visual.layer3[0].conv2.weight
    print("STEP", i, f"total={float(sum(rec.values())):.4f}", " ".join(f"{k}={float(v):.6f}" for k, v in rec.items()), f"w_l3={float(w.detach().float().norm()):.7f} w_head={float(tr.model.roi_heads.box_predictor.trans[0].weight.detach().norm()):.7f}", G.STATS["replays"], flush=True)
