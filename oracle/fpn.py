"""CPU restatement of the FPN / Swin extension (coin_amd/modeling/fpn.py, swin.py) -- TEST INFRASTRUCTURE ONLY.

**Parity unpinned.**  /root/reference contains no FPN neck, no multi-level pooler, no 2-FC head and no Swin backbone (SURVEY
finding 2: the reference is CLIP-ResNet C4 + res5, configs/coin/Base-Cloud.yaml:3-5,32-39), so there is no reference file, test or
golden vector these functions could be pinned to.  They restate the PUBLISHED algorithms (Lin et al., Feature Pyramid Networks,
2017: lateral 1x1 / output 3x3 convolutions, nearest 2x top-down pathway, RoI level k = floor(k0 + log2(sqrt(wh) / 224)); Liu et al.,
Swin Transformer, 2021: window partition, cyclic shift with region mask, relative position bias, patch merging) with deliberately
different code structure from the product (explicit loops over levels / RoIs / windows / heads) so that the product's batched
formulation is checked against an independent one.  Only tests/ may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from . import d2


def fpn_forward(feats: Dict[str, torch.Tensor], sd: Dict[str, torch.Tensor], names=("res2", "res3", "res4", "res5")) -> Dict[str, torch.Tensor]:
    """sd: state dict of the neck (fpn_lateral{2..5}.weight/bias, fpn_output{2..5}.weight/bias)."""
    inner = {}
    top = None
    for lvl in range(len(names) + 1, 1, -1):                      # 5, 4, 3, 2
        x = feats[names[lvl - 2]].double()
        lat = F.conv2d(x, sd[f"fpn_lateral{lvl}.weight"].double(), sd[f"fpn_lateral{lvl}.bias"].double())
        if top is not None:
            h, w = lat.shape[-2:]
            up = torch.zeros_like(lat)
            for yy in range(h):                                     # nearest: source index floor(dst * src / dst_size)
                for xx in range(w):
                    up[:, :, yy, xx] = top[:, :, min(yy * top.shape[-2] // h, top.shape[-2] - 1), min(xx * top.shape[-1] // w, top.shape[-1] - 1)]
            lat = lat + up
        top = lat
        inner[lvl] = lat
    out = {f"p{lvl}": F.conv2d(inner[lvl], sd[f"fpn_output{lvl}.weight"].double(), sd[f"fpn_output{lvl}.bias"].double(), padding=1) for lvl in inner}
    out[f"p{len(names) + 2}"] = out[f"p{len(names) + 1}"][:, :, ::2, ::2]   # max-pool with kernel 1, stride 2
    return out


def roi_level(box, k_min=2, k_max=5, canonical_size=224.0, k0=4) -> int:
    w, h = max(float(box[2] - box[0]), 0.0), max(float(box[3] - box[1]), 0.0)
    k = math.floor(k0 + math.log2(math.sqrt(w * h) / canonical_size + 1e-8))
    return int(min(max(k, k_min), k_max))


def multilevel_roi_align(feats: List[torch.Tensor], strides: List[int], rois: torch.Tensor, out: int = 7) -> torch.Tensor:
    """One RoI at a time on the level its size selects (torchvision roi_align, aligned, adaptive sampling: oracle.d2)."""
    res = []
    k_min = int(math.log2(strides[0]))
    for r in rois:
        k = roi_level(r[1:], k_min, k_min + len(feats) - 1)
        f = feats[k - k_min].double()
        res.append(d2.roi_align_torch(f, r.double().view(1, 5), (out, out), 1.0 / strides[k - k_min], 0, True))
    return torch.cat(res, dim=0) if res else torch.zeros(0, feats[0].shape[1], out, out, dtype=torch.float64)


def two_fc(x: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """x [R, C, 7, 7]; features enter fc1 in (h, w, c) order (the product's channels-last bytes)."""
    v = x.double().permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    h1 = torch.clamp(v @ sd["fc1.weight"].double().t() + sd["fc1.bias"].double(), min=0)
    return torch.clamp(h1 @ sd["fc2.weight"].double().t() + sd["fc2.bias"].double(), min=0)


def window_attention(qkv: np.ndarray, bias: np.ndarray, mask, heads: int, scale: float) -> np.ndarray:
    """qkv [B, T, 3*heads*hd] ([3][heads][hd] on the last axis), bias [heads, T, T], mask [nW, T, T] or None -> [B, T, heads*hd]."""
    b, t, c3 = qkv.shape
    hd = c3 // (3 * heads)
    x = qkv.astype(np.float64).reshape(b, t, 3, heads, hd)
    out = np.zeros((b, t, heads * hd))
    for w in range(b):
        for h in range(heads):
            q, k, v = x[w, :, 0, h], x[w, :, 1, h], x[w, :, 2, h]
            s = scale * (q @ k.T) + bias[h].astype(np.float64)
            if mask is not None:
                s = s + mask[w % mask.shape[0]].astype(np.float64)
            s = s - s.max(axis=1, keepdims=True)
            p = np.exp(s)
            p /= p.sum(axis=1, keepdims=True)
            out[w, :, h * hd:(h + 1) * hd] = p @ v
    return out


def window_attention_t(qkv: torch.Tensor, bias: torch.Tensor, mask, heads: int, scale: float) -> torch.Tensor:
    """The same loops as `window_attention` on float64 torch tensors (differentiable: the gradients of the Swin blocks and of the
    window-attention backward kernel are checked against autograd through THIS formulation).  mask: tensor [nW, T, T] or None."""
    b, t, c3 = qkv.shape
    hd = c3 // (3 * heads)
    x = qkv.double().reshape(b, t, 3, heads, hd)
    rows = []
    for w in range(b):
        cols = []
        for h in range(heads):
            q, k, v = x[w, :, 0, h], x[w, :, 1, h], x[w, :, 2, h]
            s = scale * (q @ k.t()) + bias[h].double()
            if mask is not None:
                s = s + mask[w % mask.shape[0]].double()
            s = s - s.max(dim=1, keepdim=True).values.detach()
            p = torch.exp(s)
            p = p / p.sum(dim=1, keepdim=True)
            cols.append(p @ v)
        rows.append(torch.cat(cols, dim=1))
    return torch.stack(rows)


def _layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w.double() + b.double()


def _region_id(y, x, hp, wp, ws, shift):
    ry = 0 if y < hp - ws else (1 if y < hp - shift else 2)
    rx = 0 if x < wp - ws else (1 if x < wp - shift else 2)
    return ry * 3 + rx


def swin_block(x: torch.Tensor, sd: Dict[str, torch.Tensor], heads: int, ws: int, shift: int, rel_index: torch.Tensor) -> torch.Tensor:
    """x [N, H, W, C] (float64).  Windows are cut one by one; the shifted variant gathers the rolled positions explicitly and masks
    token pairs that come from different regions of the un-rolled map."""
    n, h, w, c = x.shape
    t = ws * ws
    y = _layer_norm(x, sd["norm1.weight"], sd["norm1.bias"])
    hp, wp = h + (-h) % ws, w + (-w) % ws
    yp = torch.zeros(n, hp, wp, c, dtype=torch.float64)
    yp[:, :h, :w] = y
    if min(hp, wp) <= ws:
        shift = 0
    bias = sd["relative_position_bias_table"].double()[rel_index.view(-1)].view(t, t, heads).permute(2, 0, 1)
    scale = (c // heads) ** -0.5
    outp = torch.zeros_like(yp)
    for b in range(n):
        for wy in range(hp // ws):
            for wx in range(wp // ws):
                # window positions in the ROLLED map = positions (p + shift) mod size of the original one
                ys = [(wy * ws + i + shift) % hp for i in range(ws)]
                xs = [(wx * ws + j + shift) % wp for j in range(ws)]
                tok = torch.stack([yp[b, yy, xx] for yy in ys for xx in xs])                               # [T, C]
                qkv = tok @ sd["qkv.weight"].double().t() + sd["qkv.bias"].double()
                m = None
                if shift:
                    rid = np.array([_region_id(wy * ws + i, wx * ws + j, hp, wp, ws, shift) for i in range(ws) for j in range(ws)])
                    m = torch.from_numpy(np.where(rid[:, None] != rid[None, :], -100.0, 0.0))[None]
                att = window_attention_t(qkv[None], bias, m, heads, scale)[0]
                o = att @ sd["proj.weight"].double().t() + sd["proj.bias"].double()
                k = 0
                for yy in ys:
                    for xx in xs:
                        outp[b, yy, xx] = o[k]
                        k += 1
    x = x + outp[:, :h, :w]
    z = _layer_norm(x, sd["norm2.weight"], sd["norm2.bias"])
    z = z @ sd["fc1.weight"].double().t() + sd["fc1.bias"].double()
    z = 0.5 * z * (1.0 + torch.erf(z / math.sqrt(2.0)))
    return x + z @ sd["fc2.weight"].double().t() + sd["fc2.bias"].double()


def patch_merging(x: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    n, h, w, c = x.shape
    hp, wp = h + h % 2, w + w % 2
    xp = torch.zeros(n, hp, wp, c, dtype=torch.float64)
    xp[:, :h, :w] = x
    out = torch.zeros(n, hp // 2, wp // 2, 4 * c, dtype=torch.float64)
    for i in range(hp // 2):
        for j in range(wp // 2):
            out[:, i, j] = torch.cat([xp[:, 2 * i, 2 * j], xp[:, 2 * i + 1, 2 * j], xp[:, 2 * i, 2 * j + 1], xp[:, 2 * i + 1, 2 * j + 1]], dim=-1)
    out = _layer_norm(out, sd["norm.weight"], sd["norm.bias"])
    return out @ sd["reduction.weight"].double().t()


def swin_forward(image: torch.Tensor, sd: Dict[str, torch.Tensor], depths, heads, ws: int, rel_index: torch.Tensor, patch: int = 4) -> Dict[str, torch.Tensor]:
    """sd: state dict of coin_amd.modeling.swin.SwinTransformer."""
    sub = lambda prefix: {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    x = image.double()
    n, _, h, w = x.shape
    x = F.pad(x, (0, (-w) % patch, 0, (-h) % patch))
    x = F.conv2d(x, sd["patch_embed.weight"].double(), sd["patch_embed.bias"].double(), stride=patch).permute(0, 2, 3, 1)
    x = _layer_norm(x, sd["patch_norm.weight"], sd["patch_norm.bias"])
    out = {}
    for i, (d, hd) in enumerate(zip(depths, heads)):
        for k in range(d):
            x = swin_block(x, sub(f"stages.{i}.{k}."), hd, ws, 0 if k % 2 == 0 else ws // 2, rel_index)
        o = sub(f"out_norms.{i}.")
        out[f"res{i + 2}"] = _layer_norm(x, o["weight"], o["bias"]).permute(0, 3, 1, 2)
        if i < len(depths) - 1:
            x = patch_merging(x, sub(f"merges.{i}."))
    return out
