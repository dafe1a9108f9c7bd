#!/usr/bin/env python3
"""Micro-benchmark (development tool): coin_conv_gemm_bf16 against the library paths at the res5 convolution shapes of the benchmark
step (forward and data-gradient; random bf16 data; HIP events)."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import torch.nn.functional as F

from coin_amd import kernels as K

torch.backends.cudnn.benchmark = True


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


res = {}
for name, n, h, w, ci, co, ks in [("l4.0.conv1", 2048, 14, 14, 1024, 512, 1), ("l4.0.conv2", 2048, 14, 14, 512, 512, 3), ("l4.0.conv3", 2048, 7, 7, 512, 2048, 1),
                                  ("l4.0.down", 2048, 7, 7, 1024, 2048, 1), ("l4.1.conv1", 2048, 7, 7, 2048, 512, 1), ("l4.1.conv2", 2048, 7, 7, 512, 512, 3)]:
    x = torch.randn(n, ci, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, ks, ks, device="cuda", dtype=torch.bfloat16) * 0.02).contiguous(memory_format=torch.channels_last)
    m = n * h * w
    flop = 2.0 * m * ci * co * ks * ks
    pad = ks // 2
    t_lib = timeit(lambda: F.conv2d(x, wt, padding=pad))
    gy = torch.randn(n, co, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    t_lib_d = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False]))
    t_lib_w = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False]))
    xa = x.permute(0, 2, 3, 1).reshape(m, ci)
    ga = gy.permute(0, 2, 3, 1).reshape(m, co)
    wk = wt.permute(0, 2, 3, 1).reshape(co, ks * ks * ci)
    wd = wt.flip(2, 3).permute(1, 2, 3, 0).contiguous().reshape(ci, ks * ks * co)
    out = torch.empty(m, co, device="cuda", dtype=torch.bfloat16)
    gx = torch.empty(m, ci, device="cuda", dtype=torch.bfloat16)
    sp = (h, w, ci) if ks == 3 else None
    spd = (h, w, co) if ks == 3 else None
    t_f = timeit(lambda: K.conv_gemm(xa, wk, spatial=sp, out=out))
    t_fs = timeit(lambda: K.conv_gemm(xa, wk, spatial=sp, out=out, stats_rows=m))
    t_d = timeit(lambda: K.conv_gemm(ga, wd, spatial=spd, out=gx))
    t_w = timeit(lambda: K.conv_wgrad(ga, xa, spatial=sp)) if K.conv_wgrad_ok(co, ci) else float("nan")
    if K.conv_wgrad_ok(co, ci):
        wref = torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1].float()
        wgot = K.conv_wgrad(ga.contiguous(), xa.contiguous(), spatial=sp).view(co, ks, ks, ci).permute(0, 3, 1, 2)
        werr = float((wgot - wref).abs().max() / wref.abs().max())
    else:
        werr = float("nan")
    ref = F.conv2d(x, wt, padding=pad).permute(0, 2, 3, 1).reshape(m, co).float()
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    res[name] = {"M": m, "K": ks * ks * ci, "N": co, "lib_fwd_ms": t_lib, "lib_fwd_TF": flop / t_lib / 1e9, "ours_fwd_ms": t_f, "ours_fwd_TF": flop / t_f / 1e9,
                 "ours_fwd_stats_ms": t_fs, "lib_dgrad_ms": t_lib_d, "lib_dgrad_TF": flop / t_lib_d / 1e9, "ours_dgrad_ms": t_d, "ours_dgrad_TF": flop / t_d / 1e9,
                 "lib_wgrad_ms": t_lib_w, "lib_wgrad_TF": flop / t_lib_w / 1e9, "ours_wgrad_ms": t_w, "ours_wgrad_TF": flop / t_w / 1e9,
                 "wgrad_rel_err_vs_lib": werr, "rel_err_vs_lib": err}
    print(name, json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res[name].items()}), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/gemmbench.json", "w"), indent=1)
