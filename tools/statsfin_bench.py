#!/usr/bin/env python3
"""coin_conv_gemm_stats_finalize at the step's shapes (the workgroup shape was chosen with this script: 8 channels x 128 tile lanes)."""
import json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from coin_amd import kernels as K
from kbench import timeit

res = {}
for m, n in ((401408, 512), (100352, 2048), (100352, 512), (66800, 128), (16700, 1024), (16700, 256)):
    tiles = (m + 255) // 256
    part = torch.randn(tiles, 3, n, device="cuda").abs()
    rm, rv = torch.zeros(n, device="cuda"), torch.ones(n, device="cuda")
    t = timeit(lambda: K.conv_stats_finalize(part, m, n, m, 1e-5, 0.1, rm, rv), iters=50)
    res[f"{m}x{n}"] = round(t * 1e6, 2)
print(json.dumps(res))
