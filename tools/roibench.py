#!/usr/bin/env python3
"""RoIAlign forward A/B at the benchmark shape: column-walk kernel (default) vs the per-sample kernel (lab variant 1), three box-size
distributions; prints ms and algorithmic GB/s (map + boxes + output bytes)."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch

from coin_amd import _lib, kernels as K
from kbench import timeit


def boxes(n_img, per_img, lo, hi, g):
    out = []
    for i in range(n_img):
        bw = torch.rand(per_img, generator=g) * (hi - lo) + lo
        bh = torch.rand(per_img, generator=g) * (hi - lo) + lo
        x0 = torch.rand(per_img, generator=g) * (1333 - bw).clamp(min=1)
        y0 = torch.rand(per_img, generator=g) * (800 - bh).clamp(min=1)
        out.append(torch.stack([torch.full((per_img,), float(i)), x0, y0, x0 + bw, y0 + bh], 1))
    return torch.cat(out)


def main():
    g = torch.Generator().manual_seed(0)
    n, c, h, w, r = 4, 1024, 50, 83, 2048
    # the variant hook exists only in the lab build of the library (tools/build_lab.sh -> tools/lab/libcoin_hip_lab.so)
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab", "libcoin_hip_lab.so")
    lab = _lib.lib().coin_roi_align_lab_variant
    res = {}
    for dt, name in ((torch.bfloat16, "bf16"), (torch.float32, "f32")):
        feat = torch.randn(n, h, w, c, device="cuda").to(dt)
        es = feat.element_size()
        cases = [("small_16_96", 16, 96), ("bench_32_400", 32, 400), ("large_300_800", 300, 800)]
        real = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--rois=")]
        if real:   # the RoIs of a real bench step (tools/dump_rois.py)
            cases.append(("real_step", real[0], None))
        for tag, lo, hi in cases:
            rois = (torch.load(lo) if hi is None else boxes(n, 512, lo, hi, g)).cuda()
            alg = feat.numel() * es + rois.numel() * 4 + r * 196 * c * es
            row = {}
            outs = []
            for variant, vn in ((0, "cols"), (1, "per_sample"), (4, "lab_no_loads"), (8, "lab_no_stores")):
                lab(variant)
                t = timeit(lambda: K.roi_align_fwd(feat, rois, (14, 14), 1 / 16.0))
                outs.append(K.roi_align_fwd(feat, rois, (14, 14), 1 / 16.0).float())
                row[vn] = {"ms": round(t * 1e3, 4), "GBps": round(alg / t / 1e9, 1)}
            lab(0)
            row["max_abs_diff"] = float((outs[0] - outs[1]).abs().max())
            res[f"{name}_{tag}"] = row
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
