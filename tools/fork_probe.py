#!/usr/bin/env python3
"""Round-6 probe (GPU box): does a second HIP stream buy anything for the backward's weight gradients?

    python tools/fork_probe.py [iters]

Three questions, each answered eagerly (two streams + events) and as ONE captured HIP graph with a fork / join inside:
  A. res5-sized: coin_bn_bwd on [2048,7,7,2048] (HBM-bound stream) beside coin_conv_wgrad_bf16 [100352 x 2048 x 512] (MFMA-bound,
     one 128 KiB workgroup per CU) -- do the two co-reside on the CUs, i.e. is together < alone + alone?
  B. backbone-sized: a chain of small dgrad GEMMs ([16600 x 256 x 1024] etc., 65 tiles on 256 CUs) beside their weight gradients.
  C. whether a replayed graph with two branches runs them concurrently at all on this runtime.
Prints JSON lines.
"""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

from coin_amd import kernels as K

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)


def bf(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(torch.bfloat16)


def timeit(fn, n=iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


side = torch.cuda.Stream()


def forked(main_fn, side_fn):
    """main_fn on the current stream, side_fn on `side`, fork before / join after."""
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(cur)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        side_fn()
    main_fn()
    ev2 = torch.cuda.Event()
    ev2.record(side)
    cur.wait_event(ev2)


def graph_of(fn):
    g = torch.cuda.CUDAGraph()
    fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    return g


def report(tag, main_fn, side_fn):
    out = {"case": tag}
    out["main_alone_ms"] = timeit(main_fn)
    out["side_alone_ms"] = timeit(side_fn)
    out["serial_ms"] = timeit(lambda: (main_fn(), side_fn()))
    out["forked_eager_ms"] = timeit(lambda: forked(main_fn, side_fn))
    try:
        gs = graph_of(lambda: (main_fn(), side_fn()))
        out["graph_serial_ms"] = timeit(gs.replay)
        gf = graph_of(lambda: forked(main_fn, side_fn))
        out["graph_forked_ms"] = timeit(gf.replay)
    except Exception as e:  # noqa
        out["graph_error"] = f"{type(e).__name__}: {e}"[:300]
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}), flush=True)


# ---------------------------------------------------------------- A: res5 conv3's BatchNorm backward beside conv3's weight gradient
n, h, w, c = 2048, 7, 7, 2048
x = bf(n, h, w, c)
dy = bf(n, h, w, c, scale=0.05)
mean = torch.zeros(c, device=dev)
rstd = torch.ones(c, device=dev)
gamma = torch.ones(c, device=dev)
beta = torch.zeros(c, device=dev)
gy = bf(n * h * w, c, scale=0.05)       # [100352, 2048]
xin = bf(n * h * w, 512)                # [100352, 512]
report("A res5: bn_bwd [2048,7,7,2048] | wgrad [100352 x 2048 x 512]",
       lambda: K.bn_bwd(x, dy, None, mean, rstd, gamma, beta, False, 1, False),
       lambda: K.conv_wgrad(gy, xin))

# A2: a dgrad GEMM (MFMA-bound, 160 KiB of LDS per CU) beside a weight gradient (128 KiB per CU): cannot co-reside -- what does it cost?
wd = bf(512, 2048, scale=0.05)
report("A2 res5: dgrad [100352 x 512 x 2048] | wgrad [100352 x 2048 x 512]",
       lambda: K.conv_gemm(gy, wd),
       lambda: K.conv_wgrad(gy, xin))

# A3: 3x3 wgrad (long) beside bn_bwd + dgrad chain
x3 = bf(n * 14 * 14 // 4, 512)
gy3 = bf(n * 14 * 14 // 4, 512, scale=0.05)
report("A3 res5: bn_bwd + dgrad | 3x3 wgrad [100352 x 512 x 4608]",
       lambda: (K.bn_bwd(x, dy, None, mean, rstd, gamma, beta, False, 1, False), K.conv_gemm(gy, wd)),
       lambda: K.conv_wgrad(gy3, x3, spatial=(7, 7, 512)))

# ---------------------------------------------------------------- B: layer3-sized chain (res4 map of the benchmark: 4 x 50 x 83 = 16600 rows)
m = 4 * 50 * 83
g256 = bf(m, 256, scale=0.05)
g1024 = bf(m, 1024, scale=0.05)
x256 = bf(m, 256)
x1024 = bf(m, 1024)
w_a = bf(1024, 256, scale=0.05)     # dgrad of conv3: [m,1024] -> [m,256]:  gemm [16600 x 256 x 1024]
w_b = bf(256, 2304, scale=0.05)     # dgrad of conv2 3x3
w_c = bf(256, 1024, scale=0.05)     # dgrad of conv1: [m,256] -> [m,1024]
x4 = x256.view(4, 50, 83, 256)
d4 = g256.view(4, 50, 83, 256)
m256 = torch.zeros(256, device=dev)
r256 = torch.ones(256, device=dev)


def chain_main():
    for _ in range(3):
        K.conv_gemm(g1024, w_c)                                        # [m,1024] x [256,1024]^T -> [m,256]
        K.bn_bwd(x4, d4, None, m256, r256, r256, m256, True, 1, False)
        K.conv_gemm(g256, w_b, spatial=(50, 83, 256))                  # 3x3
        K.bn_bwd(x4, d4, None, m256, r256, r256, m256, True, 1, False)
        K.conv_gemm(g256, w_a)                                         # [m,256] x [1024,256]^T -> [m,1024]


def chain_side():
    for _ in range(3):
        K.conv_wgrad(g1024, x256)                        # conv3: Cout 1024 <- Cin 256
        K.conv_wgrad(g256, x256, spatial=(50, 83, 256))  # conv2 3x3
        K.conv_wgrad(g256, x1024)                        # conv1: Cout 256 <- Cin 1024


report("B layer3 x3 blocks: dgrad + bn_bwd chain | the 9 weight gradients", chain_main, chain_side)

# ---------------------------------------------------------------- C: two independent low-occupancy kernels (NMS-like): does a graph run its branches concurrently?
a = torch.randn(64, 1 << 14, device=dev)


def slow_small(t):
    for _ in range(20):
        t = torch.cumsum(t, dim=1)
    return t


report("C two low-occupancy chains (20 cumsums of [64, 16384] each)", lambda: slow_small(a), lambda: slow_small(a))
