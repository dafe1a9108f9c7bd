#!/usr/bin/env python3
"""Where does the graphed backbone stretch go wrong in the trainer?  (development probe)
1. allocator history around the capture: allocations made on the capturing stream that did NOT land in the graph's private pool;
2. per-parameter comparison of the backward replay with an eager recomputation from the same static input / output gradient, inside the
   step (eager work between the two replays) and back to back, before and after an empty_cache()."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from coin_amd import graphs as G
from coin_amd.engine import PRETrainer

torch.backends.cudnn.benchmark = True
if os.environ.get("COIN_GRAPH_PROBE") == "trace":
    G.TRACE = []
cfg = bench.build_cfg(1, "cuda:0", "bf16")
torch.manual_seed(cfg.SEED)
tr = PRETrainer(cfg)
tr.model.step_graphs = True
tr.model.roi_heads.step_graphs = os.environ.get("COIN_SEG", "backbone") == "both"

orig_capture = G.GraphedSegment._capture
SNAP = {}


def capture(self, inputs):
    if not self.name.startswith("backbone"):
        return orig_capture(self, inputs)
    torch.cuda.memory._record_memory_history(enabled="all", context="alloc", stacks="python")
    try:
        ent = orig_capture(self, inputs)
    finally:
        SNAP["snap"] = torch.cuda.memory._snapshot()
        torch.cuda.memory._record_memory_history(enabled=None)
    return ent


G.GraphedSegment._capture = capture


def analyse():
    snap = SNAP["snap"]
    segs = sorted((s["address"], s["address"] + s["total_size"], tuple(s.get("segment_pool_id", (0, 0))), s["stream"]) for s in snap["segments"])
    cap = torch.cuda.graph.default_capture_stream.cuda_stream
    print("capture stream", cap, "segments", len(segs), "pools", sorted({s[2] for s in segs}))

    def where(addr):
        for a, b, pool, st in segs:
            if a <= addr < b:
                return pool, st
        return None, None

    bad = {}
    n = 0
    for tr_ in snap["device_traces"]:
        for ev in tr_:
            if ev["action"] != "alloc":
                continue
            n += 1
            pool, st = where(ev["addr"])
            if ev["stream"] == cap and pool == (0, 0):
                fr = [f for f in ev.get("frames", []) if "/repo/" in f.get("filename", "")][:4]
                key = tuple((os.path.basename(f["filename"]), f["line"]) for f in fr)
                bad.setdefault(key, []).append(ev["size"])
            elif ev["stream"] != cap and pool not in ((0, 0), None):
                fr = [f for f in ev.get("frames", []) if "/repo/" in f.get("filename", "")][:4]
                print("  non-capture-stream alloc in a private pool:", ev["size"], pool, [(os.path.basename(f["filename"]), f["line"]) for f in fr])
    print("alloc events", n, "; capture-stream allocations in the DEFAULT pool:", sum(len(v) for v in bad.values()))
    for k, v in bad.items():
        print("   ", k, len(v), "allocs, bytes", sorted(set(v))[:6])


def compare(tag, back_to_back):
    seg = tr.model._seg_backbone[1]
    ent = next(iter(seg.graphs.values()))
    bufs = list(seg.buffers_fn())
    saved = [b.detach().clone() for b in bufs]
    if back_to_back:
        ent.fwd.replay(); ent.bwd.replay()
        with torch.no_grad():
            for b, s in zip(bufs, saved): b.copy_(s)
    torch.cuda.synchronize()
    got = [None if g is None else g.detach().float().clone() for g in ent.grads_p]
    got_out = ent.outs[0].detach().float().clone()
    x = ent.static_in[0].detach().clone()
    go = ent.static_gout[0].detach().clone()
    with torch.enable_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out = seg.fn(x)
    if (G.TRACE is not None) and G.TRACE:
        t2 = []
        ref = G._backward_on_this_thread([out], [go], list(ent.params), trace=t2)
        relt = lambda a, b: float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp(min=1e-30)) if torch.isfinite(a.float()).all() else float("inf")
        shown = 0
        print(f"[{tag}] trace nodes graph {len(G.TRACE)} eager {len(t2)}")
        for i, ((n1, a1, o1), (n2, a2, o2)) in enumerate(zip(G.TRACE, t2)):
            nm = type(n1).__name__ if not hasattr(n1, "_forward_cls") else n1._forward_cls.__name__
            nm2 = type(n2).__name__ if not hasattr(n2, "_forward_cls") else n2._forward_cls.__name__
            def saved_of(n):
                if hasattr(n, "_forward_cls"):
                    return list(n.saved_tensors)
                return [getattr(n, k) for k in dir(n) if k.startswith("_saved_") and torch.is_tensor(getattr(n, k, None))]
            ea = [relt(x, y) for x, y in zip(a1, a2) if x is not None and y is not None and x.numel()]
            eo = [relt(x, y) for x, y in zip(o1, o2) if x is not None and y is not None and x.numel()]
            try:
                es = [(relt(x, y) if (x is not None and x.numel() and x.shape == y.shape) else -1.0) for x, y in zip(saved_of(n1), saved_of(n2))]
            except Exception as e:
                es = [repr(e)[:40]]
            flag = max([0.0] + [e for e in ea + eo + es if isinstance(e, float)])
            if any(isinstance(e, float) and e > 0.05 for e in es) and "in-step" in tag:
                for kx, (ta, tb) in enumerate(zip(saved_of(n1), saved_of(n2))):
                    if ta is None or not ta.numel() or ta.shape != tb.shape:
                        continue
                    fa, fb = ta.contiguous().view(-1), tb.contiguous().view(-1)
                    d = (fa != fb) & ~(torch.isnan(fa) & torch.isnan(fb))
                    if bool(d.any()):
                        idx = d.nonzero().view(-1)
                        esz = ta.element_size()
                        print(f"      node {i} saved{kx} ptr {ta.data_ptr():#x} bytes {ta.numel() * esz}: differing {int(d.sum())} of {ta.numel()}, first byte +{int(idx[0]) * esz} last byte +{int(idx[-1]) * esz + esz};"
                              f" sample {fa[idx[:6]].float().tolist()} vs {fb[idx[:6]].float().tolist()}")
                print(f"   SAVED-BAD node {i} {nm}: saved {[('%.1e' % e) if isinstance(e, float) else e for e in es]} shapes {[tuple(t.shape) if t is not None else None for t in saved_of(n1)]}")
            if flag > 0.05 and shown < 4:
                shown += 1
                print(f"   node {i} {nm}/{nm2}: args {['%.1e' % e for e in ea]} saved {[('%.1e' % e) if isinstance(e, float) else e for e in es]} outs {['%.1e' % e for e in eo]}"
                      f" out shapes {[tuple(x.shape) for x in o1 if x is not None]}")
    else:
        ref = torch.autograd.grad(out, ent.params, go, allow_unused=True)
    with torch.no_grad():
        for b, s in zip(bufs, saved): b.copy_(s)
    names = {id(p): n for n, p in tr.model.backbone.encoder.visual.named_parameters()}
    rel = lambda a, b: float((a - b.float()).abs().max() / b.float().abs().max().clamp(min=1e-30)) if torch.isfinite(a).all() else float("inf")
    print(f"[{tag}] out err {rel(got_out, out.detach()):.2e}")
    worst = []
    for p, a, b in zip(ent.params, got, ref):
        if a is None or b is None:
            continue
        e = rel(a, b)
        worst.append((e, names[id(p)]))
    badl = [(e, n) for e, n in worst if e > 0.2]
    print(f"[{tag}] params {len(worst)}, bad (>20%) {len(badl)}: " + ", ".join(f"{n}={e:.1e}" for e, n in badl[:60]))


def classify():
    """Every tensor the captured backward touches (saved by the forward, gradient in / out of a node): which pool's segment holds it and
    is its block still allocated?"""
    snap = torch.cuda.memory_snapshot()
    blocks = []
    for sgm in snap:
        a = sgm["address"]
        for b in sgm["blocks"]:
            addr = b.get("address", a)
            blocks.append((addr, addr + b["size"], tuple(sgm.get("segment_pool_id", (0, 0))), b["state"]))
            a = addr + b["size"]
    blocks.sort()
    import bisect as _b
    starts = [b[0] for b in blocks]

    def look(t):
        i = _b.bisect_right(starts, t.data_ptr()) - 1
        if i < 0 or not (blocks[i][0] <= t.data_ptr() < blocks[i][1]):
            return (None, "unmapped")
        return blocks[i][2], blocks[i][3]

    seen = {}
    for i, (n, args, outs) in enumerate(G.TRACE):
        nm = type(n).__name__ if not hasattr(n, "_forward_cls") else n._forward_cls.__name__
        items = []
        if hasattr(n, "_forward_cls"):
            try:
                items += [("saved%d" % k, t) for k, t in enumerate(n.saved_tensors) if t is not None]
            except Exception as e:
                items += []
            items += [("ctx." + k, v) for k, v in vars(n).items() if torch.is_tensor(v)] if hasattr(n, "__dict__") else []
        else:
            items += [(k, getattr(n, k)) for k in dir(n) if k.startswith("_saved_") and torch.is_tensor(getattr(n, k, None))]
        items += [("arg%d" % k, t) for k, t in enumerate(args) if t is not None]
        items += [("out%d" % k, t) for k, t in enumerate(outs) if t is not None]
        for k, t in items:
            if not t.is_cuda or t.numel() == 0:
                continue
            pool, state = look(t)
            if pool != (0, 1) or state != "active_allocated":
                key = (i, nm, k)
                print(f"   node {i} {nm}.{k}: shape {tuple(t.shape)} {t.dtype} ptr {t.data_ptr():#x} pool {pool} state {state}")
    print("classified", len(G.TRACE), "nodes")


steps = 0
real_step = tr.optimizer.step
MODE = {"check": False}


def stage_check(tag):
    seg = tr.model._seg_backbone[1]
    ent = next(iter(seg.graphs.values()))
    bufs = list(seg.buffers_fn()); keep = [b.detach().clone() for b in bufs]
    n0 = G.TRACE[0][0]
    def show(what):
        torch.cuda.synchronize()
        sv = n0.saved_tensors
        print(f"   [{tag}] {what}: " + " ".join(f"saved{k}:{'ok' if bool(torch.isfinite(t.float()).all()) else 'BAD'}({float(t.float().abs().max()):.2e})" for k, t in enumerate(sv) if t is not None)
              + f" out:{'ok' if bool(torch.isfinite(ent.outs[0].float()).all()) else 'BAD'}", flush=True)
    show("before anything")
    ent.fwd.replay(); show("after fwd.replay")
    ent.bwd.replay(); show("after bwd.replay")
    print(f"   [{tag}] grads finite: {sum(bool(torch.isfinite(g).all()) for g in ent.grads_p if g is not None)} of {sum(g is not None for g in ent.grads_p)}")
    with torch.no_grad():
        for b, s_ in zip(bufs, keep): b.copy_(s_)


def step(*a, **k):
    if MODE["check"]:
        if (G.TRACE is not None) and steps == 4:
            n0 = G.TRACE[0][0]
            for kk, t in enumerate(n0.saved_tensors):
                if t is not None:
                    print(f"   node0 saved{kk} {tuple(t.shape)} {t.dtype} finite={bool(torch.isfinite(t.float()).all())} absmax={float(t.float().abs().max()):.3e} ptr={t.data_ptr():#x}")
            classify()
        compare(f"step {steps} in-step", False)
        compare(f"step {steps} back-to-back", True)
    return real_step(*a, **k)  # noqa


tr.optimizer.step = step
for i in range(4):
    steps = i
    MODE["check"] = i == 3
    rec = tr.run_step()
    print("STEP", i, f"total={float(sum(rec.values())):.4f}", G.STATS["captures"], G.STATS["replays"], flush=True)
analyse()
if (G.TRACE is not None):
    classify()
    stage_check("pre-empty")
torch.cuda.synchronize(); torch.cuda.empty_cache()
print("emptied the cache", flush=True)
if (G.TRACE is not None):
    classify()
    vis = tr.model.backbone.encoder.visual
    from coin_amd import layers as L
    stage_check("post-empty")
    print("live shadow of layer3.5.conv3", hex(L._SHADOWS[id(vis.layer3[5].conv3.weight)].tensor.data_ptr()))
for i in range(4, 7):
    steps = i
    MODE["check"] = True
    rec = tr.run_step()
    print("STEP", i, f"total={float(sum(rec.values())):.4f}", G.STATS["captures"], G.STATS["replays"], flush=True)
