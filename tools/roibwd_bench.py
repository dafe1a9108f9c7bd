#!/usr/bin/env python3
"""RoIAlign backward at the benchmark shape ([4,50,83,1024], 2048 boxes, 14 x 14 bins): the tile shapes of the gather kernel (lab build:
`coin_roi_align_lab_bwd_cfg`) on the RoIs of a real bench step (profiles/r4_real_rois.pt) and on synthetic box-size distributions;
prints ms, algorithmic TB/s (gradient tile + boxes + map bytes) and whether the map equals the first configuration's bit for bit
(0 = what the product library ships, 10 = round 2's kernel, the others: see BWD_LAB_CASES in roi_align.hip).
  python tools/roibwd_bench.py [cfg ...]"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..")))
import torch

from coin_amd import _lib

_lib.LIB_PATH = os.path.join(HERE, "lab", "libcoin_hip_lab.so")
from coin_amd import kernels as K
from roibench import boxes


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    cfgs = [int(a) for a in sys.argv[1:] if a.lstrip("-").isdigit()] or [10, 0, 20, 22, 34, 37]
    lab = _lib.lib().coin_roi_align_lab_bwd_cfg
    g = torch.Generator().manual_seed(0)
    n, c, h, w, r = 4, 1024, 50, 83, 2048
    res = {}
    real = os.path.join(HERE, "..", "profiles", "r4_real_rois.pt")
    cases = [("real_step", torch.load(real)), ("bench_32_400", boxes(n, 512, 32, 400, g)), ("large_300_800", boxes(n, 512, 300, 800, g))]
    if os.environ.get("ROIBWD_QUICK"):   # the real step's boxes in bf16 only (debug-switch sweeps: cfg | dbg << 8)
        cases = cases[:1]
    for dt, name in ((torch.bfloat16, "bf16"), (torch.float32, "f32"))[: 1 if os.environ.get("ROIBWD_QUICK") else 2]:
        go = torch.randn(r, 14, 14, c, device="cuda").to(dt)
        for tag, rois in cases:
            rois = rois.cuda().float()
            alg = go.numel() * go.element_size() + rois.numel() * 4 + n * h * w * c * 4
            ref = None
            alg1 = None   # configuration 20's map: the summation order of the ring kernels (30+), which must equal it bit for bit
            for cfg in cfgs:
                lab(cfg)
                out = K.roi_align_bwd(go, rois, (n, h, w, c), 1 / 16.0)
                if ref is None:
                    ref = out.clone()
                if cfg == 20:
                    alg1 = out.clone()
                t = timeit(lambda: K.roi_align_bwd(go, rois, (n, h, w, c), 1 / 16.0))
                res[f"{name}_{tag}_cfg{cfg}"] = {"ms": round(t * 1e3, 4), "TBps": round(alg / t / 1e12, 3), "bit_equal_cfg0": bool(torch.equal(out, ref)),
                                                 "max_rel_diff": float((out - ref).abs().max() / ref.abs().max()),
                                                 "bit_equal_cfg20": None if alg1 is None else bool(torch.equal(out, alg1))}
                print(f"{name}_{tag}_cfg{cfg}", res[f"{name}_{tag}_cfg{cfg}"], flush=True)
            lab(0)
    json.dump(res, open(os.path.join(HERE, "..", "gpurun_out", "roibwd_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
