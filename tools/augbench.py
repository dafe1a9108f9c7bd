#!/usr/bin/env python3
"""Micro-benchmark (development tool): the coin_aug_* kernels at a Cityscapes-shaped input (1024x2048 -> 600x1200) against Pillow on one
host thread, per operation and for a whole strong + weak pair.  Run on the GPU box:  python tools/augbench.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image, ImageEnhance, ImageFilter, ImageOps

from coin_amd import kernels as K


def gpu_ms(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def cpu_ms(fn, iters=5):
    fn()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    return (time.perf_counter() - t) / iters * 1e3


a = np.random.default_rng(0).integers(0, 256, (1024, 2048, 3), dtype=np.uint8)
d = torch.from_numpy(a).cuda()
pil = Image.fromarray(a, "RGB")
res = {}
res["resize 1024x2048->600x1200"] = {"gpu_ms": gpu_ms(lambda: K.aug_resize_bilinear(d, 600, 1200, True)), "pillow_ms": cpu_ms(lambda: pil.resize((1200, 600), Image.BILINEAR)),
                                     "alg_MB": (a.nbytes + 600 * 1200 * 3) / 1e6}
w = K.aug_resize_bilinear(d, 600, 1200)
pw = pil.resize((1200, 600), Image.BILINEAR)
mb = 2 * 600 * 1200 * 3 / 1e6
for name, g, c in [
    ("brightness", lambda: K.aug_point_op(w, K.AUG_BRIGHTNESS, fparam=1.2), lambda: ImageEnhance.Brightness(pw).enhance(1.2)),
    ("contrast", lambda: K.aug_point_op(w, K.AUG_CONTRAST, fparam=1.2), lambda: ImageEnhance.Contrast(pw).enhance(1.2)),
    ("saturation", lambda: K.aug_point_op(w, K.AUG_SATURATION, fparam=1.2), lambda: ImageEnhance.Color(pw).enhance(1.2)),
    ("hue", lambda: K.aug_point_op(w, K.AUG_HUE, iparam=12), lambda: pw.convert("HSV").convert("RGB")),
    ("grayscale", lambda: K.aug_point_op(w, K.AUG_GRAYSCALE), lambda: pw.convert("L").convert("RGB")),
    ("solarize", lambda: K.aug_point_op(w, K.AUG_SOLARIZE, iparam=128), lambda: ImageOps.solarize(pw, 128)),
    ("gaussian blur r=1.5", lambda: K.aug_gaussian_blur(w, 1.5), lambda: pw.filter(ImageFilter.GaussianBlur(1.5))),
    ("to CHW", lambda: K.aug_point_op(w, K.AUG_COPY, out_chw=True), lambda: np.ascontiguousarray(np.asarray(pw).transpose(2, 0, 1))),
]:
    res[name] = {"gpu_ms": gpu_ms(g), "pillow_ms": cpu_ms(c), "alg_MB": mb * (6 if "blur" in name else 1)}
for v in res.values():
    v["gpu_GBps"] = v["alg_MB"] / v["gpu_ms"]


def pair():
    x = K.aug_resize_bilinear(d, 600, 1200, True)
    s = x
    for op, kw in ((K.AUG_BRIGHTNESS, {"fparam": 1.2}), (K.AUG_CONTRAST, {"fparam": 0.8}), (K.AUG_SATURATION, {"fparam": 1.3}), (K.AUG_HUE, {"iparam": 12})):
        s = K.aug_point_op(s, op, **kw)
    s = K.aug_gaussian_blur(s, 1.5)
    return K.aug_point_op(s, K.AUG_COPY, out_chw=True), K.aug_point_op(x, K.AUG_COPY, out_chw=True)


def pair_pil():
    x = pil.resize((1200, 600), Image.BILINEAR)
    x = Image.fromarray(np.ascontiguousarray(np.flip(np.asarray(x), axis=1)), "RGB")
    s = ImageEnhance.Color(ImageEnhance.Contrast(ImageEnhance.Brightness(x).enhance(1.2)).enhance(0.8)).enhance(1.3)
    s = s.convert("HSV").convert("RGB").filter(ImageFilter.GaussianBlur(1.5))
    return np.asarray(s).transpose(2, 0, 1).copy(), np.asarray(x).transpose(2, 0, 1).copy()


g, c = gpu_ms(pair), cpu_ms(pair_pil, 3)
res["strong+weak pair (jitter x4 + blur)"] = {"gpu_ms": g, "pillow_ms": c, "gpu_pairs_per_s": 1e3 / g, "pillow_pairs_per_s_one_thread": 1e3 / c}
print(json.dumps(res, indent=1))
