#!/usr/bin/env python3
"""Are the library's convolutions safe inside a captured HIP graph on this stack?  (development probe)
Captures single library convolutions (the shapes the backbone stretch left on the library), replays them bare, after unrelated host /
device allocations and after empty_cache(), compares with the eager call; dumps the graph's node types."""
import os, re, sys, collections
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = os.environ.get("BENCHMARK", "1") == "1"
dev = "cuda"
gen = torch.Generator(device=dev).manual_seed(5)


def rnd(*shape):
    return torch.randn(*shape, device=dev, generator=gen).to(torch.bfloat16)


def node_types(g, tag):
    path = f"/tmp/graph_{tag}.dot"
    try:
        g.debug_dump(path)
        txt = open(path).read()
        kinds = collections.Counter(re.findall(r'label="[^"]*?(KERNEL|MEMCPY|MEMSET|HOST|EMPTY|EVENT|MEM_ALLOC|MEM_FREE|CHILD)', txt, flags=re.I))
        names = collections.Counter(re.findall(r'label="[^"\\]*?\\n?([A-Za-z_][A-Za-z0-9_:<>]*)', txt))
        return dict(kinds), len(txt), txt
    except Exception as e:
        return repr(e), 0, ""


def case(tag, make_inputs, fn):
    ins = make_inputs()
    for _ in range(3):
        ref = fn(*ins)
    torch.cuda.synchronize()
    ref = [r.clone() for r in (ref if isinstance(ref, (tuple, list)) else (ref,))]
    g = torch.cuda.CUDAGraph()
    g.enable_debug_mode()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = fn(*ins)
    out = out if isinstance(out, (tuple, list)) else (out,)
    kinds, n, txt = node_types(g, tag)

    def err():
        torch.cuda.synchronize()
        return max(float((o.float() - r.float()).abs().max() / r.float().abs().max().clamp(min=1e-30)) if torch.isfinite(o.float()).all() else float("inf")
                   for o, r in zip(out, ref))

    g.replay(); e1 = err()
    for o in out:
        o.fill_(float("nan"))
    junk_host = [bytearray(os.urandom(1 << 20)) for _ in range(64)]     # churn the host heap
    junk = [torch.full((1 << 26,), float("nan"), device=dev) for _ in range(8)]
    g.replay(); e2 = err()
    del junk
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    junk = [torch.full((1 << 26,), float("nan"), device=dev) for _ in range(16)]
    for o in out:
        o.fill_(float("nan"))
    g.replay(); e3 = err()
    print(f"MIOPEN-GRAPH {tag:34s} replay {e1:.2e} | after host+device churn {e2:.2e} | after empty_cache {e3:.2e} | nodes {kinds}", flush=True)
    if os.environ.get("DUMP") == "1":
        print(txt[:3000])
    del junk


cl = torch.channels_last
# forward 1x1, 256 -> 1024 on the res4 map (layer3.x.conv3)
case("fwd 1x1 256->1024 @50x84", lambda: (rnd(4, 256, 50, 84).contiguous(memory_format=cl), rnd(1024, 256, 1, 1).contiguous(memory_format=cl)),
     lambda x, w: F.conv2d(x, w))
# forward 3x3 on the stem-like map (frozen stages of the teacher)
case("fwd 3x3 64->64 @200x336", lambda: (rnd(4, 64, 200, 336).contiguous(memory_format=cl), rnd(64, 64, 3, 3).contiguous(memory_format=cl)),
     lambda x, w: F.conv2d(x, w, padding=1))
# backward-weights 1x1, 512 -> 128 on the res3 map (layer2.x.conv1)
case("wrw 1x1 512->128 @100x168", lambda: (rnd(4, 128, 100, 168).contiguous(memory_format=cl), rnd(4, 512, 100, 168).contiguous(memory_format=cl),
                                          rnd(128, 512, 1, 1).contiguous(memory_format=cl)),
     lambda gy, x, w: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1])
# backward-weights 3x3, 128 -> 128
case("wrw 3x3 128->128 @100x168", lambda: (rnd(4, 128, 100, 168).contiguous(memory_format=cl), rnd(4, 128, 100, 168).contiguous(memory_format=cl),
                                          rnd(128, 128, 3, 3).contiguous(memory_format=cl)),
     lambda gy, x, w: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1])
# backward-data + weights 1x1 1024 -> 256 (layer3.x.conv1 below the GEMM threshold)
case("bwd 1x1 1024->256 @50x84", lambda: (rnd(4, 256, 50, 84).contiguous(memory_format=cl), rnd(4, 1024, 50, 84).contiguous(memory_format=cl),
                                         rnd(256, 1024, 1, 1).contiguous(memory_format=cl)),
     lambda gy, x, w: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, True, False])[:2])
