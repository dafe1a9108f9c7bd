"""CPU restatement of the CLIP-teacher relabelling used to collect pre-training targets -- TEST INFRASTRUCTURE ONLY.

Follows coin/modeling/utils.py:93-125 (AttentionPool2d), coin/modeling/roi_heads/clip_roi_heads.py:19-87 (CLIPRes5ROIHeads) and
coin/modeling/meta_arch/clip_rcnn.py:87-151 (CLIP.forward / get_clip_result / preprocess_image / preprocess_boxes).
Pinned by tests/test_oracle_golden.py against tests/golden/clip_relabel.npz (captured from the reference's modules).
"""
from __future__ import annotations

import copy
import math
from typing import Dict, List

import torch
import torch.nn.functional as F
from torch import nn

from . import d2


class AttentionPool2d(nn.Module):
    """CLIP's attention pooling: the mean token queries all HW+1 tokens once; same parameter names as the reference."""

    def __init__(self, spacial_dim: int, embed_dim: int, num_heads: int, output_dim: int = None):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.randn(spacial_dim ** 2 + 1, embed_dim) / embed_dim ** 0.5)
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.c_proj = nn.Linear(embed_dim, output_dim or embed_dim)
        self.num_heads = num_heads

    def forward(self, x):
        n, c = x.shape[:2]
        tok = x.flatten(2).transpose(1, 2)                                   # [N, HW, C]
        tok = torch.cat([tok.mean(dim=1, keepdim=True), tok], dim=1) + self.positional_embedding.to(x.dtype)
        hd = c // self.num_heads
        q = self.q_proj(tok[:, :1]).view(n, 1, self.num_heads, hd).transpose(1, 2) / math.sqrt(hd)
        k = self.k_proj(tok).view(n, -1, self.num_heads, hd).transpose(1, 2)
        v = self.v_proj(tok).view(n, -1, self.num_heads, hd).transpose(1, 2)
        att = torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v          # [N, heads, 1, hd]
        return self.c_proj(att.transpose(1, 2).reshape(n, c))


def clip_probs(region_feats: torch.Tensor, text_feats: torch.Tensor, logit_scale: torch.Tensor) -> torch.Tensor:
    """clip_roi_heads.py:79-85: softmax(exp(logit_scale) * cos(region, text))."""
    f = region_feats / region_feats.norm(dim=1, keepdim=True)
    t = text_feats / text_feats.norm(dim=1, keepdim=True)
    return (logit_scale.exp() * f @ t.t()).softmax(dim=-1)


@torch.no_grad()
def clip_relabel(backbone, attnpool, text_feats, logit_scale, pixel_mean, pixel_std, batched_input: Dict, pre_result: Dict,
                 pooler_resolution: int = 14, scale: float = 1.0 / 16) -> Dict:
    """CLIP.forward (clip_rcnn.py:87-132) for one image: re-score every cloud box with CLIP, drop those labelled background.
    `backbone`: module with ``forward(x)['res4']`` and ``.layer4``."""
    img = batched_input["image"]
    mean = torch.tensor(pixel_mean).view(3, 1, 1)
    std = torch.tensor(pixel_std).view(3, 1, 1)
    x = ((img.float() / 255.0) - mean) / std                                # ToTensor + Normalize
    feats = backbone(x.unsqueeze(0))["res4"]
    net_h, net_w = img.shape[1:]
    out = copy.deepcopy(pre_result)

    def process(name):
        inst = out[name]["instances"]
        if len(inst) == 0:
            return inst
        boxes = inst.pred_boxes.tensor.clone()
        boxes[:, 0::2] *= net_w / batched_input["width"]
        boxes[:, 1::2] *= net_h / batched_input["height"]
        rois = torch.cat([boxes.new_zeros(len(boxes), 1), boxes], dim=1)
        tiles = d2.roi_align_torch(feats, rois, (pooler_resolution, pooler_resolution), scale, 0, True)
        probs = clip_probs(attnpool(backbone.layer4(tiles)), text_feats, logit_scale)
        score, label = probs.max(1)
        new = d2.Instances(inst.image_size)
        new.pred_boxes = inst.pred_boxes
        new.pred_classes, new.scores, new.probs = label, score, probs
        return new[label != probs.size(1) - 1]

    for tag in ("RCNN", "RPN", "RPN_AUG"):
        if tag in out:
            out[tag] = {"instances": process(tag)}
    return out
