"""CLIP ModifiedResNet backbone (C4) for the MI355X path.

Mirrors coin/modeling/backbone/clip_backbone.py:150-287 (``CLIP_IMAGE``, registered builder
``build_clip_image_backbone``) and coin/modeling/utils.py:26-292 (``Bottleneck``, ``ModifiedResNet``):
identical module tree and state-dict keys (``encoder.visual.*``), ``layer4`` kept for the RoI head.

MI355X execution choices (values unchanged):
  * activations are channels-last (NHWC bytes) end to end, so RoIAlign reads coalesced channel vectors and
    res5 consumes its output without a layout change;
  * the frozen prefix (stem + layer1 at FREEZE_AT=2) runs under no_grad with FrozenBatchNorm folded into the
    convolution (w*scale, shift as bias) - no activations are kept for a backward that never happens;
  * convolutions go through torch (MIOpen) in the compute dtype; BatchNorm of the trainable stages uses
    batch statistics per GPU exactly like the reference (no SyncBN).
Weights are random-initialised as CLIP does (bn3.weight = 0) - there is no network to download RN50.pt.
"""
from __future__ import annotations

import weakref
from collections import OrderedDict
from typing import Dict

import torch
import torch.nn.functional as F
from torch import nn

from .. import layers as L
from ..registry import BACKBONE_REGISTRY
from ..structures import ShapeSpec


class FrozenBatchNorm2d(nn.Module):
    """detectron2 FrozenBatchNorm2d: y = x * w/sqrt(var+eps) + (b - mean * w/sqrt(var+eps)); buffers only."""

    def __init__(self, num_features: int, eps: float = 1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def scale_shift(self):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        return scale, self.bias - self.running_mean * scale

    def forward(self, x):
        s, b = self.scale_shift()
        return x * s.reshape(1, -1, 1, 1).to(x.dtype) + b.reshape(1, -1, 1, 1).to(x.dtype)

    @classmethod
    def convert(cls, module: nn.Module) -> nn.Module:
        if isinstance(module, (nn.BatchNorm2d, nn.SyncBatchNorm)):
            res = cls(module.num_features, module.eps)
            res.weight.data = module.weight.data.clone().detach()
            res.bias.data = module.bias.data.clone().detach()
            res.running_mean.data = module.running_mean.data
            res.running_var.data = module.running_var.data
            return res
        for name, child in module.named_children():
            new = cls.convert(child)
            if new is not child:
                module.add_module(name, new)
        return module


def _conv_bn(x, conv: nn.Conv2d, bn: nn.Module, relu: bool, residual=None, pool: int = 1):
    """conv -> norm (+ identity) (-> relu) (-> 2x2 average pool).  In the bf16 throughput mode a frozen norm and the whole
    elementwise tail are one fused pass over the convolution's output (coin_amd.layers.frozen_bn_act) -- or, where that kernel's
    thread mapping does not fit the width, the norm is folded into the convolution; the fp32 parity mode keeps the reference's
    operation order (conv, then x*scale + shift)."""
    if isinstance(bn, FrozenBatchNorm2d) and L.frozen_bn_fusable(x, conv.out_channels):
        return L.frozen_bn_act(L.conv2d(x, conv), bn, relu, residual, pool)
    if residual is not None or pool != 1:
        y = _conv_bn(x, conv, bn, False)
        y = y + residual if residual is not None else y
        y = F.relu(y) if relu else y
        return F.avg_pool2d(y, pool) if pool != 1 else y
    if isinstance(bn, FrozenBatchNorm2d) and x.dtype != torch.float32:
        w, b = _folded(conv, bn, x.dtype)
        y = F.conv2d(x, w, b, conv.stride, conv.padding)
    else:
        y = bn(L.conv2d(x, conv))
    return F.relu(y) if relu else y


_FOLDED: dict = {}
L._DERIVED_CACHES.append(_FOLDED)   # dropped by L.invalidate_storage when the EMA kernel rewrites a source tensor in place


def _folded(conv: nn.Conv2d, bn: "FrozenBatchNorm2d", dtype: torch.dtype):
    """Frozen conv + frozen norm folded into one (weight, bias) pair in the compute dtype; both are constants of the
    training run, so the pair is computed once and re-derived only if one of the five source tensors is written."""
    src = (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = tuple((t.data_ptr(), t._version) for t in src) + (dtype,)
    hit = _FOLDED.get(id(conv))
    if hit is not None and hit[0] == key and hit[1]() is conv:
        return hit[2], hit[3]
    with torch.no_grad():
        s, b = bn.scale_shift()
        w = (conv.weight * s.view(-1, 1, 1, 1)).to(dtype).contiguous(memory_format=torch.channels_last)
        b = b.to(dtype)
    _FOLDED[id(conv)] = (key, weakref.ref(conv, lambda _r, k=id(conv): _FOLDED.pop(k, None)), w, b)
    return w, b


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1):
        super().__init__()
        out = planes * self.expansion
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.avgpool = nn.AvgPool2d(stride) if stride > 1 else nn.Identity()
        self.conv3 = nn.Conv2d(planes, out, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(out)
        self.relu = nn.ReLU(inplace=True)
        self.stride = stride
        self.downsample = None
        if stride > 1 or inplanes != out:
            self.downsample = nn.Sequential(OrderedDict([
                ("-1", nn.AvgPool2d(stride)), ("0", nn.Conv2d(inplanes, out, 1, stride=1, bias=False)), ("1", nn.BatchNorm2d(out))]))

    def forward(self, x, mean_pool: bool = False):
        """mean_pool: return the spatial mean [N,C,1,1] of the block output instead of the output (the RoI head's
        `x.mean(dim=[2,3])`, clip_roi_heads.py:207-208, folded into bn3 + identity + ReLU: the activation is never stored)."""
        if mean_pool and isinstance(self.bn1, FrozenBatchNorm2d):
            return self.forward(x).mean(dim=[2, 3], keepdim=True)
        if isinstance(self.bn1, FrozenBatchNorm2d):  # frozen stage (layer1 at FREEZE_AT=2): plain inference ops
            y = _conv_bn(x, self.conv1, self.bn1, True)
            y = _conv_bn(y, self.conv2, self.bn2, True, pool=self.stride)
            if self.downsample is not None:
                x = _conv_bn(self.downsample[0](x), self.downsample[1], self.downsample[2], False)
            return _conv_bn(y, self.conv3, self.bn3, True, residual=x)
        # trainable stage: every BatchNorm + elementwise tail is a fused HIP stream (coin_amd.layers.bn_act)
        pool = 2 if self.stride > 1 else 1
        assert self.stride in (1, 2)
        # x: the identity branch's tap (gradient fan-in fused); for the stride-2 block the tap comes back already average-pooled
        fork = "pool" if (self.downsample is not None and pool == 2) else True
        y, x = L.conv_bn_act(x, self.conv1, self.bn1, relu=True, fork=fork)
        y = L.conv_bn_act(y, self.conv2, self.bn2, relu=True, pool=pool)               # ReLU and the anti-aliasing avg-pool fused in
        if self.downsample is not None:
            sx = L.conv_bn_act(x, self.downsample[1], self.downsample[2], relu=False)
        else:
            sx = x
        return L.conv_bn_act(y, self.conv3, self.bn3, relu=True, residual=sx, pool=0 if mean_pool else 1)  # bn3 + identity + ReLU


class ModifiedResNet(nn.Module):
    def __init__(self, layers, width: int = 64, out_features=("res4",), freeze_at: int = 0):
        super().__init__()
        self.conv1 = nn.Conv2d(3, width // 2, 3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(width // 2)
        self.conv2 = nn.Conv2d(width // 2, width // 2, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width // 2)
        self.conv3 = nn.Conv2d(width // 2, width, 3, padding=1, bias=False)
        self.bn3 = nn.BatchNorm2d(width)
        self.avgpool = nn.AvgPool2d(2)
        self._inplanes = width
        self.layer1 = self._make_layer(width, layers[0])
        self.layer2 = self._make_layer(width * 2, layers[1], stride=2)
        self.layer3 = self._make_layer(width * 4, layers[2], stride=2)
        self.layer4 = self._make_layer(width * 8, layers[3], stride=2)  # used by the RoI head (C4)
        self._out_features = list(out_features)
        self._out_feature_channels = {"stem": width, "res2": width * 4, "res3": width * 8, "res4": width * 16, "res5": width * 32}
        self._out_feature_strides = {"stem": 4, "res2": 4, "res3": 8, "res4": 16, "res5": 32}
        self.freeze_at = freeze_at
        self.freeze(freeze_at)

    def _make_layer(self, planes, blocks, stride=1):
        mods = [Bottleneck(self._inplanes, planes, stride)]
        self._inplanes = planes * Bottleneck.expansion
        mods += [Bottleneck(self._inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def freeze(self, freeze_at: int = 0):
        """coin/modeling/utils.py:243-284."""
        def fz(m):
            for p in m.parameters():
                p.requires_grad = False
            return FrozenBatchNorm2d.convert(m)

        self.freeze_at = max(int(freeze_at), int(getattr(self, "freeze_at", 0)))
        if freeze_at >= 1:
            for name in ("conv1", "bn1", "conv2", "bn2", "conv3", "bn3"):
                setattr(self, name, fz(getattr(self, name)))
        for idx, stage in enumerate([self.layer1, self.layer2, self.layer3, self.layer4], start=2):
            if freeze_at >= idx:
                for block in stage.children():
                    fz(block)
        return self

    def _stem(self, x):
        x = _conv_bn(x, self.conv1, self.bn1, True)
        x = _conv_bn(x, self.conv2, self.bn2, True)
        return _conv_bn(x, self.conv3, self.bn3, True, pool=2)   # self.avgpool = nn.AvgPool2d(2) fused into the last pass

    def forward(self, x, frozen_done: bool = False) -> Dict[str, torch.Tensor]:
        """frozen_done: `x` is already the output of `frozen_forward` (the stages without trainable parameters)."""
        assert x.dim() == 4
        frozen_prefix = min(self.freeze_at, 4)  # stages [1, freeze_at] have no trainable parameter
        stages = [self._stem, self.layer1, self.layer2, self.layer3]
        for i, stage in enumerate(stages, start=1):
            if i <= frozen_prefix and frozen_done:
                continue
            if i <= frozen_prefix and not x.requires_grad:
                with torch.no_grad():
                    x = stage(x)
            else:
                x = stage(x)
        return {"res4": x}

    def forward_stages(self, x, first: int, last: int) -> torch.Tensor:
        """Stages first .. last (1 = stem, 2 = layer1, 3 = layer2, 4 = layer3) applied to x, the frozen ones without a graph -- the
        piecewise form of `forward` (the trainers capture layer3 as a HIP graph and run the stages before it eagerly)."""
        frozen_prefix = min(self.freeze_at, 4)
        stages = [self._stem, self.layer1, self.layer2, self.layer3]
        for i in range(first, last + 1):
            if i <= frozen_prefix and not x.requires_grad:
                with torch.no_grad():
                    x = stages[i - 1](x)
            else:
                x = stages[i - 1](x)
        return x

    def forward_pyramid(self, x) -> Dict[str, torch.Tensor]:
        """res2 .. res5 (strides 4, 8, 16, 32): the bottom-up pathway of the FPN extension (coin_amd/modeling/fpn.py); layer4 runs on
        the whole map here instead of on RoI tiles."""
        frozen_prefix = min(self.freeze_at, 5)
        out = {}
        stages = [("stem", self._stem), ("res2", self.layer1), ("res3", self.layer2), ("res4", self.layer3), ("res5", self.layer4)]
        for i, (name, stage) in enumerate(stages, start=1):
            if i <= frozen_prefix and not x.requires_grad:
                with torch.no_grad():
                    x = stage(x)
            else:
                x = stage(x)
            out[name] = x
        return out

    @torch.no_grad()
    def frozen_forward(self, x) -> torch.Tensor:
        """Stem + the frozen stages only.  Their result depends on the images alone (never on a weight update), so a trainer
        may compute it for the NEXT batch while the current step is still running."""
        stages = [self._stem, self.layer1, self.layer2, self.layer3]
        for stage in stages[: min(self.freeze_at, 4)]:
            x = stage(x)
        return x

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n]) for n in self._out_features}

    @staticmethod
    def res4_hw(h: int, w: int):
        """Spatial size of res4 for an [h, w] input: stride-2 3x3 conv (pad 1), then three floor-halving average pools."""
        f = lambda v: ((v + 1) // 2) // 2 // 2 // 2
        return f(int(h)), f(int(w))


class AttentionPool2d(nn.Module):
    """CLIP's attention pooling (coin/modeling/utils.py:93-125): the mean token queries the HW + 1 position-embedded tokens once
    (multi-head attention with separate q / k / v projections); same parameter names as the reference.  Used by the CLIP teacher
    that relabels the cloud detector's boxes (clip_rcnn.py:87-132); the mean-pool detectors delete it."""

    def __init__(self, spacial_dim: int, embed_dim: int, num_heads: int, output_dim: int = None):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.randn(spacial_dim ** 2 + 1, embed_dim) / embed_dim ** 0.5)
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.c_proj = nn.Linear(embed_dim, output_dim or embed_dim)
        self.num_heads = num_heads

    def forward(self, x):  # [N, C, H, W] (any memory format) -> [N, output_dim]
        n, c = x.shape[:2]
        tok = x.flatten(2).transpose(1, 2)
        tok = torch.cat([tok.mean(dim=1, keepdim=True), tok], dim=1) + self.positional_embedding.to(x.dtype)
        hd = c // self.num_heads
        split = lambda t: t.view(n, -1, self.num_heads, hd).transpose(1, 2)
        q = split(L.linear(tok[:, :1], self.q_proj.weight, self.q_proj.bias))
        k = split(L.linear(tok, self.k_proj.weight, self.k_proj.bias))
        v = split(L.linear(tok, self.v_proj.weight, self.v_proj.bias))
        att = F.scaled_dot_product_attention(q, k, v)                        # [N, heads, 1, hd]
        return L.linear(att.transpose(1, 2).reshape(n, c), self.c_proj.weight, self.c_proj.bias)


class _ImageEncoder(nn.Module):
    def __init__(self, visual):
        super().__init__()
        self.visual = visual
        self.attnpool = None  # meanpool configs delete it (clip_rcnn.py:226-227)


_ARCH = {"RN50": ((3, 4, 6, 3), 64, 1024), "RN101": ((3, 4, 23, 3), 64, 512), "RN50x4": ((4, 6, 10, 6), 80, 640)}


class CLIP_IMAGE(nn.Module):
    size_divisibility = 0

    def __init__(self, type: str = "RN50", out_features=("res4",), freeze_at: int = 2, update_backbone: bool = True,
                 layers=None, width=None, attnpool_dim: int = None, attnpool_heads: int = None):
        super().__init__()
        l, w, _ = _ARCH.get(type, _ARCH["RN50"])
        layers, width = layers or l, width or w
        self.type = type
        self.encoder = _ImageEncoder(ModifiedResNet(layers, width, out_features, freeze_at))
        self._width = width
        if attnpool_dim:
            self.add_attnpool(attnpool_dim, attnpool_heads)
        for name, p in self.encoder.visual.named_parameters():  # clip_backbone.py:56-61
            if name.endswith("bn3.weight") and name.startswith("layer"):
                nn.init.zeros_(p)
        self.update_backbone = update_backbone
        if not update_backbone:  # clip_backbone.py:211-217
            for n, p in self.encoder.visual.named_parameters():
                if "layer4" not in n:
                    p.requires_grad = False

    @classmethod
    def from_config(cls, cfg):
        a = cfg.AMD.ARCH
        from .text_encoder import text_dim_of

        keep_attnpool = cfg.MODEL.ROI_HEADS.POOLING_TYPE == "attnpool"
        return cls(type=cfg.MODEL.TEACHER_OFFLINE.TYPE or "RN50", out_features=cfg.MODEL.RESNETS.OUT_FEATURES,
                   freeze_at=cfg.MODEL.BACKBONE.FREEZE_AT, update_backbone=cfg.CLOUD.UPDATE_BACKBONE,
                   layers=tuple(a.LAYERS) or None, width=a.WIDTH or None, attnpool_dim=text_dim_of(cfg) if keep_attnpool else None)

    layer4 = property(lambda self: self.encoder.visual.layer4)
    attnpool = property(lambda self: self.encoder.attnpool)

    def add_attnpool(self, output_dim: int, heads: int = None):
        """clip_backbone.py:52,62-67 (image_resolution 224 -> 7x7 tokens); only the CLIP relabelling teacher keeps it."""
        c = self._width * 32
        self.encoder.attnpool = AttentionPool2d(7, c, heads or max(c // 64, 1), output_dim)
        for lin in (self.encoder.attnpool.q_proj, self.encoder.attnpool.k_proj, self.encoder.attnpool.v_proj, self.encoder.attnpool.c_proj):
            nn.init.normal_(lin.weight, std=c ** -0.5)
        return self

    def del_attnpool(self):
        self.encoder.attnpool = None

    def output_shape(self):
        return self.encoder.visual.output_shape()

    def train(self, mode: bool = True):  # clip_backbone.py:223-234
        if self.update_backbone:
            return super().train(mode)
        self.training = False
        self.encoder.training = False
        for m in self.encoder.children():
            if m is not None:
                m.eval()
        self.encoder.visual.layer4.train(mode)
        return self

    def forward(self, image: torch.Tensor, frozen_done: bool = False):
        return self.encoder.visual(image, frozen_done=frozen_done)

    def frozen_forward(self, image: torch.Tensor) -> torch.Tensor:
        return self.encoder.visual.frozen_forward(image)


@BACKBONE_REGISTRY.register()
def build_clip_image_backbone(cfg, input_shape=None):
    return CLIP_IMAGE.from_config(cfg)
