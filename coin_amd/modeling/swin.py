"""Swin-T bottom-up network for the FPN extension (BASELINE.json configs[4]; **no counterpart in /root/reference**, SURVEY finding 2).

Follows the published Swin Transformer (patch 4, window 7, shifted windows, relative position bias, depths (2, 2, 6, 2), heads
(3, 6, 12, 24), head dimension 32) and returns {"res2" .. "res5"} at strides 4 .. 32 for coin_amd/modeling/fpn.py.  Validated
against this repository's own restatement (oracle/fpn.py): parity unpinned by construction.

MI355X path: activations stay [N, H, W, C] (channels-last is the natural layout of every LayerNorm / Linear here); in the bf16
throughput mode the attention of all windows x heads of a block is ONE launch of ``coin_window_attn_fwd`` (QK^T, bias + shift
mask, softmax and PV on MFMA, csrc/window_attn.hip); its backward is ``coin_window_attn_bwd`` (dQ, dK, dV on MFMA and the gradient
of the trained relative-position-bias table, summed over the windows in a fixed order).  fp32 / CPU runs use the torch formulation.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import kernels as K
from .. import layers as L


def window_attention_reference(qkv: torch.Tensor, bias: torch.Tensor, mask: Optional[torch.Tensor], heads: int, scale: float) -> torch.Tensor:
    """softmax(scale q k^T + bias + mask) v.  qkv [B, T, 3*C] ([3][heads][hd] along the last axis), bias [heads, T, T],
    mask [nW, T, T] or None (window b uses mask[b % nW]) -> [B, T, C]."""
    b, t, c3 = qkv.shape
    hd = c3 // (3 * heads)
    q, k, v = qkv.view(b, t, 3, heads, hd).permute(2, 0, 3, 1, 4).float()
    att = (q * scale) @ k.transpose(-1, -2) + bias.float().unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        att = (att.view(b // nw, nw, heads, t, t) + mask.float().view(1, nw, 1, t, t)).view(b, heads, t, t)
    return (att.softmax(dim=-1) @ v).transpose(1, 2).reshape(b, t, heads * hd).to(qkv.dtype)


def _pad64(x: torch.Tensor, fill_cols: float) -> torch.Tensor:
    """[.., T, T] -> [.., 64, 64]: new columns = fill_cols (masked keys), new rows = 0."""
    t = x.shape[-1]
    out = x.new_zeros(x.shape[:-2] + (64, 64))
    out[..., :, t:] = fill_cols
    out[..., :t, :t] = x
    return out.contiguous()


class _WindowAttn(Function):
    @staticmethod
    def forward(ctx, qkv, bias, mask, mask64, heads, scale):
        out = K.window_attn_fwd(qkv, _pad64(bias.detach().float(), -1e30), mask64, heads, scale)
        ctx.save_for_backward(qkv, bias, mask if mask is not None else qkv.new_zeros(0))
        ctx.heads, ctx.scale, ctx.has_mask, ctx.mask64 = heads, scale, mask is not None, mask64
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        qkv, bias, mask = ctx.saved_tensors
        # coin_window_attn_bwd: P recomputed with the forward's arithmetic, dP / dV / dQ / dK on MFMA, d bias summed over the windows in
        # a fixed order (round 3 recomputed the attention with differentiable torch ops here: ~25 launches and fp32 [B, heads, T, T] tensors)
        gq, gb = K.window_attn_bwd(qkv, _pad64(bias.detach().float(), -1e30), ctx.mask64, dout.contiguous().to(torch.bfloat16), ctx.heads, ctx.scale)
        return gq, gb.to(bias.dtype), None, None, None, None


def window_attention(qkv, bias, mask, mask64, heads: int, scale: float) -> torch.Tensor:
    if qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.shape[-1] == 3 * heads * 32 and qkv.shape[1] <= 64:
        return _WindowAttn.apply(qkv.contiguous(), bias, mask, mask64, heads, scale)
    return window_attention_reference(qkv, bias, mask, heads, scale)


def _relative_position_index(ws: int) -> torch.Tensor:
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def shift_mask(hp: int, wp: int, ws: int, shift: int, device) -> torch.Tensor:
    """Additive attention mask of the shifted-window blocks: [nW, T, T], 0 inside a region, -100 across regions."""
    img = torch.zeros(1, hp, wp, 1)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    win = img.view(1, hp // ws, ws, wp // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
    m = win.unsqueeze(1) - win.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0).to(device)


class SwinBlock(nn.Module):
    def __init__(self, dim: int, heads: int, window: int, shift: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.dim, self.heads, self.window, self.shift = dim, heads, window, shift
        self.norm1 = nn.LayerNorm(dim)
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window - 1) ** 2, heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.register_buffer("relative_position_index", _relative_position_index(window), persistent=False)
        self.norm2 = nn.LayerNorm(dim)
        self.fc1, self.fc2 = nn.Linear(dim, int(dim * mlp_ratio)), nn.Linear(int(dim * mlp_ratio), dim)
        self.scale = (dim // heads) ** -0.5
        self._masks: Dict[Tuple, Tuple[torch.Tensor, torch.Tensor]] = {}

    def _mask(self, hp, wp, device):
        key = (hp, wp, str(device))
        if key not in self._masks:
            m = shift_mask(hp, wp, self.window, self.shift, device)
            self._masks[key] = (m, _pad64(m, 0.0))  # padded keys are already removed by the bias columns
        return self._masks[key]

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # [N, H, W, C]
        n, h, w, c = x.shape
        ws, t = self.window, self.window * self.window
        y = self.norm1(x)
        ph, pw = (-h) % ws, (-w) % ws
        if ph or pw:
            y = F.pad(y, (0, 0, 0, pw, 0, ph))
        hp, wp = h + ph, w + pw
        shift = self.shift if min(hp, wp) > ws else 0
        if shift:
            y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
        win = y.view(n, hp // ws, ws, wp // ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, t, c)
        qkv = L.linear(win, self.qkv.weight, self.qkv.bias)
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(t, t, -1).permute(2, 0, 1)
        mask, mask64 = self._mask(hp, wp, x.device) if shift else (None, None)
        att = window_attention(qkv, bias, mask, mask64, self.heads, self.scale)
        att = L.linear(att, self.proj.weight, self.proj.bias)
        y = att.view(n, hp // ws, wp // ws, ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(n, hp, wp, c)
        if shift:
            y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
        if ph or pw:
            y = y[:, :h, :w, :]
        x = x + y
        return x + L.linear(F.gelu(L.linear(self.norm2(x), self.fc1.weight, self.fc1.bias)), self.fc2.weight, self.fc2.bias)


class PatchMerging(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.norm = nn.LayerNorm(4 * dim)
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)

    def forward(self, x):  # [N, H, W, C] -> [N, ceil(H/2), ceil(W/2), 2C]
        n, h, w, c = x.shape
        if h % 2 or w % 2:
            x = F.pad(x, (0, 0, 0, w % 2, 0, h % 2))
        x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], dim=-1)
        return L.linear(self.norm(x), self.reduction.weight, None)


class SwinTransformer(nn.Module):
    def __init__(self, embed_dim: int = 96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size: int = 7, patch: int = 4):
        super().__init__()
        self.patch_embed = nn.Conv2d(3, embed_dim, patch, stride=patch)
        self.patch_norm = nn.LayerNorm(embed_dim)
        self.stages, self.merges, self.out_norms = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for i, (d, hds) in enumerate(zip(depths, num_heads)):
            dim = embed_dim * 2 ** i
            self.stages.append(nn.Sequential(*[SwinBlock(dim, hds, window_size, 0 if k % 2 == 0 else window_size // 2) for k in range(d)]))
            self.merges.append(PatchMerging(dim) if i < len(depths) - 1 else nn.Identity())
            self.out_norms.append(nn.LayerNorm(dim))
        self.patch = patch
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, image: torch.Tensor) -> Dict[str, torch.Tensor]:
        n, _, h, w = image.shape
        p = self.patch
        if h % p or w % p:
            image = F.pad(image, (0, (-w) % p, 0, (-h) % p))
        x = L.conv2d(image, self.patch_embed).permute(0, 2, 3, 1)   # channels-last activations: this permute is a view
        x = self.patch_norm(x)
        out = {}
        for i, (stage, merge, norm) in enumerate(zip(self.stages, self.merges, self.out_norms)):
            x = stage(x)
            out[f"res{i + 2}"] = norm(x).permute(0, 3, 1, 2)       # logical NCHW, channels-last memory: what the FPN convolutions take
            x = merge(x)
        return out
