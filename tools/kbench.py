#!/usr/bin/env python3
"""Micro-benchmarks of the hand-written kernels at BASELINE sizes (HIP events on the launch stream)."""
import json
import sys
import os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from coin_amd import kernels as K


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def rois_like_bench(n_img, per_img, g):
    out = []
    for i in range(n_img):
        bw = torch.rand(per_img, generator=g) * 368 + 32
        bh = torch.rand(per_img, generator=g) * 368 + 32
        x0 = torch.rand(per_img, generator=g) * (1333 - bw)
        y0 = torch.rand(per_img, generator=g) * (800 - bh)
        out.append(torch.stack([torch.full((per_img,), float(i)), x0, y0, x0 + bw, y0 + bh], 1))
    return torch.cat(out)


def main():
    g = torch.Generator().manual_seed(0)
    res = {}
    n, c, h, w, r = 4, 1024, 50, 83, 2048
    real = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--rois=")]   # e.g. profiles/r4_real_rois.pt (tools/dump_rois.py)
    rois = (torch.load(real[0]) if real else rois_like_bench(n, 512, g)).cuda()
    for dt, name in ((torch.bfloat16, "bf16"), (torch.float32, "f32")):
        feat = torch.randn(n, h, w, c, device="cuda").to(dt)
        es = feat.element_size()
        t = timeit(lambda: K.roi_align_fwd(feat, rois, (14, 14), 1 / 16.0))
        alg = feat.numel() * es + rois.numel() * 4 + r * 196 * c * es
        res[f"roi_align_fwd_{name}"] = {"ms": t * 1e3, "alg_GB": alg / 1e9, "GBps": alg / t / 1e9}
        go = torch.randn(r, 14, 14, c, device="cuda").to(dt)
        gf = torch.zeros(n, h, w, c, device="cuda")
        t = timeit(lambda: K.roi_align_bwd(go, rois, (n, h, w, c), 1 / 16.0, grad_feat=gf))
        alg = go.numel() * es + rois.numel() * 4 + gf.numel() * 4
        res[f"roi_align_bwd_{name}"] = {"ms": t * 1e3, "alg_GB": alg / 1e9, "GBps": alg / t / 1e9}
    for (m, nn, k) in [(2048, 1024, 2048), (2048, 2048, 1024), (8192, 8192, 8192), (401408, 512, 1024), (100352, 2048, 512)]:
        a = (torch.randn(m, k, device="cuda") * 0.5).to(torch.bfloat16)
        b = (torch.randn(nn, k, device="cuda") * 0.05).to(torch.bfloat16)
        out = torch.empty(m, nn, device="cuda", dtype=torch.bfloat16)
        t = timeit(lambda: K.gemm_nt(a, b, out=out))
        t2 = timeit(lambda: torch.matmul(a, b.t()))
        res[f"gemm_bf16_{m}x{nn}x{k}"] = {"ms": t * 1e3, "TFLOPs": 2 * m * nn * k / t / 1e12, "torch_ms": t2 * 1e3,
                                         "torch_TFLOPs": 2 * m * nn * k / t2 / 1e12}
    a = torch.randn(2048, 2048, device="cuda")
    b = torch.randn(1024, 2048, device="cuda")
    t = timeit(lambda: K.gemm_nt(a, b))
    res["gemm_f32_2048x1024x2048"] = {"ms": t * 1e3, "TFLOPs": 2 * 2048 * 1024 * 2048 / t / 1e12}
    x = torch.randn(2048, 9, device="cuda") * 10
    lab = torch.randint(0, 9, (2048,), device="cuda")
    t = timeit(lambda: K.mil_ce(x, labels=lab, avg_positives=True))
    res["mil_ce_2048x9"] = {"us": t * 1e6}
    print(json.dumps(res, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/kbench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
