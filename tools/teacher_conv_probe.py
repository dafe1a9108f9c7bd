#!/usr/bin/env python3
"""Which library convolutions of the teacher's inference pass are not replay-safe?  (development probe)
Records every F.conv2d configuration of one eager `_inference_core` pass of the bf16 RN50 detector, then captures each one alone in a HIP
graph and replays it after unrelated allocations / empty_cache (tools/miopen_graph_probe.py's protocol)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
from coin_amd.config import get_cfg
from coin_amd.engine import PRETrainer

torch.backends.cudnn.benchmark = os.environ.get("BENCHMARK", "1") == "1"
cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs", "coin", "PRETRAINS", "CLIPDET_synthetic.yaml"))
cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", 1, "AMD.SYNTHETIC.NUM_IMAGES", 1, "AMD.COMPUTE_DTYPE", "bf16", "AMD.TEXT_TEMPLATES", 2,
                     "MODEL.DEVICE", "cuda:0", "AMD.STEP_GRAPHS", False, "AMD.SYNTHETIC.HEIGHT", 608, "AMD.SYNTHETIC.WIDTH", 800])
torch.manual_seed(7)
tr = PRETrainer(cfg)
model = tr.model.eval()
strong, weak = next(tr._data_loader_iter)
batch = [dict(d) for d in weak]
seen = {}
real = F.conv2d


def spy(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    key = (tuple(x.shape), tuple(x.stride()), x.dtype, tuple(w.shape), tuple(w.stride()), b is not None, tuple(stride) if not isinstance(stride, int) else (stride,) * 2,
           tuple(padding) if not isinstance(padding, int) else (padding,) * 2)
    seen.setdefault(key, 0)
    seen[key] += 1
    return real(x, w, b, stride, padding, dilation, groups)


F.conv2d = spy
torch.nn.functional.conv2d = spy
with torch.no_grad():
    model.inference_begin(batch, branch="test")
    model._begun = None
F.conv2d = real
torch.nn.functional.conv2d = real
print("library convolution configurations in the pass:", len(seen), flush=True)
gen = torch.Generator(device="cuda").manual_seed(1)
for key, cnt in seen.items():
    xs, xst, dt, ws, wst, has_b, stride, pad = key
    x = torch.empty_strided(xs, xst, dtype=dt, device="cuda"); x.copy_(torch.randn(xs, device="cuda", generator=gen))
    w = torch.empty_strided(ws, wst, dtype=dt, device="cuda"); w.copy_(torch.randn(ws, device="cuda", generator=gen) * 0.05)
    b = (torch.randn(ws[0], device="cuda", generator=gen) * 0.1).to(dt) if has_b else None
    with torch.no_grad():
        for _ in range(3):
            ref = real(x, w, b, stride, pad)
        torch.cuda.synchronize()
        ref = ref.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = real(x, w, b, stride, pad)

        def err():
            torch.cuda.synchronize()
            return float((out.float() - ref.float()).abs().max() / ref.float().abs().max().clamp(min=1e-30)) if torch.isfinite(out.float()).all() else float("inf")

        g.replay(); e1 = err()
        out.fill_(float("nan"))
        junk = [torch.full((1 << 26,), float("nan"), device="cuda") for _ in range(8)]
        g.replay(); e2 = err()
        del junk
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        junk = [torch.full((1 << 26,), float("nan"), device="cuda") for _ in range(16)]
        out.fill_(float("nan"))
        g.replay(); e3 = err()
        del junk
    flag = "UNSAFE" if max(e1, e2, e3) > 1e-3 else "ok"
    print(f"TEACHER-CONV x{cnt} in {xs} w {ws} bias {has_b} stride {stride} pad {pad}: {e1:.1e} | {e2:.1e} | {e3:.1e}  {flag}", flush=True)
