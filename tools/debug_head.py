#!/usr/bin/env python3
"""Development tool: where does the D=1024 head's trans.0 gradient lose precision?  (run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import real_width as RW
from e2e_util import _inst
from coin_amd import layers as L
from coin_amd import kernels as K

DEV = "cuda:0"
# 1. the layer alone at the failing shape
g = torch.Generator().manual_seed(0)
x = torch.randn(512, 2048, generator=g).abs().to(DEV).requires_grad_(True)
w = (torch.randn(1024, 2048, generator=g) * 0.03).to(DEV).requires_grad_(True)
b = (torch.randn(1024, generator=g) * 0.1).to(DEV).requires_grad_(True)
dy = (torch.randn(512, 1024, generator=g) * 1e-4).to(DEV)
y = L.linear_act(x, w, b, L.ACT_LEAKY_RELU, 0.01)
y.backward(dy)
x64, w64, b64 = (t.detach().double().requires_grad_(True) for t in (x, w, b))
y64 = torch.nn.functional.leaky_relu(x64 @ w64.T + b64, 0.01)
y64.backward(dy.double())
rel = lambda a, e: float((a.double() - e).abs().max() / e.abs().max())
print("layer alone: y %.2e dx %.2e dw %.2e db %.2e" % (rel(y, y64), rel(x.grad, x64.grad), rel(w.grad, w64.grad), rel(b.grad, b64.grad)))

# 2. inside the head
z, xh = RW.head_inputs()
bp = RW.fill_head(RW.product_head(), z)
rec = {}
orig = L.linear_act
calls = []
def spy(x, weight, bias, act=L.ACT_NONE, alpha=0.01, **kw):
    out = orig(x, weight, bias, act, alpha, **kw)
    if out.requires_grad:
        i = len(calls)
        calls.append((x.detach(), weight, bias, act, out.detach()))
        out.register_hook(lambda gr, i=i: rec.__setitem__(i, gr.detach().clone()))
    return out
L.linear_act = spy
import coin_amd.modeling.fast_rcnn as FR
got = RW.run_head(bp, z, xh, _inst, device=DEV)
L.linear_act = orig
for i, (xi, wi, bi, act, out) in enumerate(calls):
    if i not in rec:
        continue
    dh = rec[i].double()
    dz = dh * torch.where(out > 0, 1.0, 0.01).double() if act == L.ACT_LEAKY_RELU else dh
    dw = dz.T @ xi.double()
    db = dz.sum(0)
    name = [n for n, p in bp.named_parameters() if p is wi]
    print(i, name, tuple(xi.shape), "->", tuple(out.shape), "grad dtype", rec[i].dtype, "x dtype", xi.dtype,
          "| dW vs fp64-of-own-operands %.2e" % rel(wi.grad, dw), "| db %.2e" % (rel(bi.grad, db) if bi is not None and bi.grad is not None else -1))

# 3. the same gradients in the fp64 oracle, row by row
from golden_util import instances
ob = RW.fill_head(RW.oracle_head(), z)
ocap = {}
for nm, mod in (("trans.0.weight", ob.trans[0]), ("trans.2.weight", ob.trans[2]), ("trans.4.weight", ob.trans[4]), ("cls_score.weight", ob.cls_score),
                ("bbox_pred.weight", ob.bbox_pred)):
    mod.register_full_backward_hook(lambda m, gi, go, nm=nm: ocap.__setitem__(nm, go[0].detach().clone()))
ex = RW.run_head(ob, z, xh, instances, dtype=torch.float64)
for i, (xi, wi, bi, act, out) in enumerate(calls):
    if i not in rec:
        continue
    name = [n for n, p in bp.named_parameters() if p is wi][0]
    dh = rec[i].double().cpu()
    dz = dh * torch.where(out.cpu() > 0, 1.0, 0.01).double() if act == L.ACT_LEAKY_RELU else dh
    e = ocap[name]
    err = (dz - e).abs()
    rowerr = err.max(1).values / e.abs().max()
    top = rowerr.topk(5)
    print(name, "dz max err / max|dz| %.2e" % float(err.max() / e.abs().max()), "| worst rows", top.indices.tolist(), ["%.1e" % v for v in top.values.tolist()],
          "| rows > 1e-5:", int((rowerr > 1e-5).sum()), "| column-sum err %.2e" % float(((dz - e).sum(0)).abs().max() / e.sum(0).abs().max()))
