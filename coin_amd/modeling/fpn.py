"""FPN extension of the adaptation-training path (BASELINE.json configs[2]-[4]): **no counterpart in the reference**.

`/root/reference` trains a CLIP-ResNet **C4** detector with a res5 RoI head (configs/coin/Base-Cloud.yaml:3-5,32-39; SURVEY finding 2):
there is no FPN neck, no multi-level pooler, no 2-FC box head and no Swin student anywhere in it.  BASELINE.json nevertheless names
"ResNet-50-FPN / ResNet-101-FPN / Swin-T-FPN" students, so these are built as NEW registry entries next to the reference's names and
validated against this repository's own CPU restatement (oracle/fpn.py) -- **parity unpinned** by construction:

  * ``build_clip_resnet_fpn_backbone``  CLIP ModifiedResNet (RN50 / RN101, `MODEL.TEACHER_OFFLINE.TYPE`) bottom-up, res2..res5 -> FPN p2..p6
  * ``build_swint_fpn_backbone``        Swin-T bottom-up (coin_amd/modeling/swin.py: window attention on MFMA) -> FPN p2..p6
  * ``OpenVocabularyFPNROIHeads``       multi-level 7x7 RoIAlign + 2-FC head in front of the reference's FastRCNNOutputLayers
                                         (text-embedding classifier, KD / MIL / box losses: unchanged, fast_rcnn.py:116-752)
  * ``DualTeacherRPN`` runs on p2..p6 with one shared head; the levels' anchors form ONE anchor set (rpn.py), so labelling,
    losses, top-k and NMS are the reference's single-level code applied to the union.
The design follows the published FPN / Faster R-CNN recipe (lateral 1x1 + output 3x3 convolutions, nearest 2x top-down pathway, p6 by
a stride-2 max-pool of p5, RoI level = floor(4 + log2(sqrt(area) / 224)) clamped to [2, 5]).
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F
from torch import nn

from .. import layers as L
from .._lib import ACT_RELU
from ..box_ops import Matcher
from ..registry import BACKBONE_REGISTRY, ROI_HEADS_REGISTRY
from ..structures import ShapeSpec
from .backbone import _ARCH, ModifiedResNet
from .fast_rcnn import FastRCNNOutputLayers
from .roi_heads import OpenVocabularyRes5ROIHeads
from .text_encoder import build_text_encoder


class FPN(nn.Module):
    """Feature pyramid over a bottom-up network that returns {name: map} (strides 4, 8, 16, 32) -> {"p2" .. "p5", "p6"}."""

    def __init__(self, in_features: List[str], in_channels: List[int], out_channels: int = 256):
        super().__init__()
        self.in_features, self.out_channels = list(in_features), out_channels
        for i, c in enumerate(in_channels, start=2):
            lat, out = nn.Conv2d(c, out_channels, 1), nn.Conv2d(out_channels, out_channels, 3, padding=1)
            for m in (lat, out):  # c2_xavier_fill
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)
            self.add_module(f"fpn_lateral{i}", lat)
            self.add_module(f"fpn_output{i}", out)

    def forward(self, feats: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        n = len(self.in_features)
        out, prev = {}, None
        for i in range(n - 1, -1, -1):  # top-down
            x = feats[self.in_features[i]]
            lat = L.conv2d(x, getattr(self, f"fpn_lateral{i + 2}"))
            if prev is not None:
                up = F.interpolate(prev, size=lat.shape[-2:], mode="nearest")
                lat = lat + up
            prev = lat
            out[f"p{i + 2}"] = L.conv2d(lat, getattr(self, f"fpn_output{i + 2}"))
        out[f"p{n + 2}"] = F.max_pool2d(out[f"p{n + 1}"], kernel_size=1, stride=2)  # LastLevelMaxPool
        return out


class _FPNBackbone(nn.Module):
    """Backbone interface of OpenVocabularyRCNN (meta_arch.py): forward(image) -> {name: map}, output_shape(), size_divisibility,
    and the `layer4` / `attnpool` slots of the C4 design (empty here: the RoI head owns its own 2-FC head)."""
    size_divisibility = 32
    layer4 = None
    attnpool = None

    def del_attnpool(self):
        pass

    def output_shape(self):
        return {f"p{i}": ShapeSpec(channels=self.fpn.out_channels, stride=2 ** i) for i in range(2, 7)}

    def forward(self, image: torch.Tensor, frozen_done: bool = False):
        return self.fpn(self.bottom_up_features(image))


class CLIPResNetFPN(_FPNBackbone):
    def __init__(self, type: str = "RN50", freeze_at: int = 2, layers=None, width=None, out_channels: int = 256):
        super().__init__()
        l, w, _ = _ARCH.get(type, _ARCH["RN50"])
        layers, width = layers or l, width or w
        self.type = type
        self.bottom_up = ModifiedResNet(layers, width, ("res2", "res3", "res4", "res5"), freeze_at)
        for name, p in self.bottom_up.named_parameters():  # CLIP's zero-init of the last norm of every block (clip_backbone.py:56-61)
            if name.endswith("bn3.weight") and name.startswith("layer"):
                nn.init.zeros_(p)
        self.fpn = FPN(["res2", "res3", "res4", "res5"], [width * 4, width * 8, width * 16, width * 32], out_channels)

    def bottom_up_features(self, image):
        return self.bottom_up.forward_pyramid(image)


@BACKBONE_REGISTRY.register()
def build_clip_resnet_fpn_backbone(cfg, input_shape=None):
    a = cfg.AMD.ARCH
    return CLIPResNetFPN(type=cfg.MODEL.TEACHER_OFFLINE.TYPE or "RN50", freeze_at=cfg.MODEL.BACKBONE.FREEZE_AT, layers=tuple(a.LAYERS) or None,
                         width=a.WIDTH or None, out_channels=cfg.MODEL.FPN.OUT_CHANNELS)


class SwinFPN(_FPNBackbone):
    def __init__(self, embed_dim: int = 96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window: int = 7, out_channels: int = 256):
        super().__init__()
        from .swin import SwinTransformer

        self.bottom_up = SwinTransformer(embed_dim=embed_dim, depths=depths, num_heads=num_heads, window_size=window)
        self.fpn = FPN(["res2", "res3", "res4", "res5"], [embed_dim * 2 ** i for i in range(4)], out_channels)

    def bottom_up_features(self, image):
        return self.bottom_up(image)


@BACKBONE_REGISTRY.register()
def build_swint_fpn_backbone(cfg, input_shape=None):
    s = cfg.MODEL.SWIN
    return SwinFPN(embed_dim=s.EMBED_DIM, depths=tuple(s.DEPTHS), num_heads=tuple(s.NUM_HEADS), window=s.WINDOW_SIZE, out_channels=cfg.MODEL.FPN.OUT_CHANNELS)


def assign_levels(boxes: torch.Tensor, min_level: int = 2, max_level: int = 5, canonical_size: float = 224.0, canonical_level: int = 4):
    """RoI -> pyramid level index (0 = min_level): floor(canonical_level + log2(sqrt(area) / canonical_size)), clamped."""
    area = (boxes[:, 2] - boxes[:, 0]).clamp(min=0) * (boxes[:, 3] - boxes[:, 1]).clamp(min=0)
    lvl = torch.floor(canonical_level + torch.log2(torch.sqrt(area) / canonical_size + 1e-8))
    return (lvl.clamp(min=min_level, max=max_level) - min_level).to(torch.int64)


class MultiLevelROIPooler(nn.Module):
    """7x7 RoIAlign (aligned, adaptive sampling) of every RoI on the pyramid level its size selects.  Sync-free (no `nonzero`, no
    data-dependent shape -- the same policy as the sync-free samplers): on the device the level index of each RoI is an operand of
    ONE launch (coin_roi_align_fwd_levels; backward: one filtered gather launch per level); the host / mixed-dtype formulation below
    pools every level for all rows and masks."""

    def __init__(self, output_size: int, scales, sampling_ratio: int = 0, min_level: int = 2):
        super().__init__()
        self.output_size, self.scales, self.sampling_ratio = (output_size, output_size), [float(s) for s in scales], int(sampling_ratio)
        self.min_level = min_level

    def forward(self, feats: List[torch.Tensor], rois: torch.Tensor) -> torch.Tensor:
        lvl = assign_levels(rois[:, 1:], self.min_level, self.min_level + len(feats) - 1)
        if rois.is_cuda and len(feats) <= 4 and all(f.dtype == feats[0].dtype for f in feats):
            # the level index rides into the kernel: every RoI is pooled ONCE, on its own level (no per-level pass over all rows, no masks)
            return L.roi_align_levels(feats, self.scales, rois, lvl, self.output_size, self.sampling_ratio, True)
        out = None
        for i, (f, s) in enumerate(zip(feats, self.scales)):
            x = L.roi_align(f, rois, self.output_size, s, self.sampling_ratio, True)
            x = x * (lvl == i).to(x.dtype).view(-1, 1, 1, 1)
            out = x if out is None else out + x
        return out


class TwoFCHead(nn.Module):
    """FastRCNNConvFCHead with two fully connected layers (+ ReLU), on the MFMA GEMM of the box head (coin_gemm_nt)."""

    def __init__(self, in_dim: int, fc_dim: int = 1024):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(in_dim, fc_dim), nn.Linear(fc_dim, fc_dim)
        for m in (self.fc1, self.fc2):  # c2_xavier_fill
            nn.init.kaiming_uniform_(m.weight, a=1)
            nn.init.constant_(m.bias, 0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1) if x.dim() == 4 else x   # channels-last bytes: (h, w, c) feature order
        if x.is_cuda:
            x = L.linear_act(x, self.fc1.weight, self.fc1.bias, ACT_RELU)
            return L.linear_act(x, self.fc2.weight, self.fc2.bias, ACT_RELU)
        return F.relu(F.linear(F.relu(F.linear(x, self.fc1.weight, self.fc1.bias)), self.fc2.weight, self.fc2.bias))


@ROI_HEADS_REGISTRY.register()
class OpenVocabularyFPNROIHeads(OpenVocabularyRes5ROIHeads):
    """The reference's RoI head logic (sampling into (fg, bg) / (A, B, bg), the C-box pass, inference; clip_roi_heads.py:90-399) with
    the RoI feature extractor replaced: multi-level 7x7 RoIAlign on p2..p5 -> 2-FC head -> [R, 1024]."""

    @classmethod
    def from_config(cls, cfg, input_shape, backgroud):
        in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
        assert len(in_features) > 1, "the FPN head pools from several levels (MODEL.ROI_HEADS.IN_FEATURES = [p2, p3, p4, p5])"
        text_encoder = build_text_encoder(cfg, backgroud)
        res, fc = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION, cfg.MODEL.ROI_BOX_HEAD.FC_DIM
        ch = input_shape[in_features[0]].channels
        head = cls(
            in_features=in_features,
            pooler=MultiLevelROIPooler(res, [1.0 / input_shape[f].stride for f in in_features], cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO,
                                       min_level=int(math.log2(input_shape[in_features[0]].stride))),
            box_predictor=FastRCNNOutputLayers.from_config(cfg, text_encoder, ShapeSpec(channels=fc, height=1, width=1)),
            pooling_type="fpn2fc",
            num_classes=len(text_encoder.classes) - 1 if backgroud else len(text_encoder.classes),
            batch_size_per_image=cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE, positive_fraction=cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION,
            proposal_matcher=Matcher(cfg.MODEL.ROI_HEADS.IOU_THRESHOLDS, cfg.MODEL.ROI_HEADS.IOU_LABELS, allow_low_quality_matches=False),
            proposal_append_gt=cfg.MODEL.ROI_HEADS.PROPOSAL_APPEND_GT, BG_TRAIN=cfg.CLOUD.BG_TRAIN)._set(
                inference_rois_per_image=cfg.MODEL.RPN.POST_NMS_TOPK_TEST)
        head.box_head = TwoFCHead(ch * res * res, fc)
        return head

    def _pooled_rows(self, features, rois, res5, attnpool):
        x = self.pooler([features[f] for f in self.in_features], rois)
        return self.box_head(x).to(self.compute_dtype)
