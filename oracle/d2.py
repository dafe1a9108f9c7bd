"""Restatement of the third-party arithmetic under the COIN hot path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PARITY UNPINNED: the
reference pins detectron2==0.5, torchvision==0.10.1+cu111, fvcore==0.1.5
(/root/reference/docs/Environment.md:8,41,102) but vendors none of them and
the build container has neither the packages nor network access.  What follows
restates their *published* behaviour for exactly the call sites the reference
uses (listed per function); ``tests/test_oracle_d2.py`` pins each function with
hand-computed known-answer vectors.

Everything is plain torch on CPU (fp32 unless the caller passes fp64); the
RoIAlign reference is a direct per-sample loop in numpy so that it shares no
code with either the torch composite or the HIP kernel it checks.
"""
from __future__ import annotations

import itertools
import math
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------- #
# structures: Boxes / Instances / ImageList
# (reference call sites: clip_roi_heads.py:8,187,302; rpn.py:10,138; clip_rcnn.py:26,297)
# --------------------------------------------------------------------------- #
class Boxes:
    """N x 4 float boxes, (x0, y0, x1, y1) absolute pixels."""

    def __init__(self, tensor: torch.Tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        tensor = tensor.to(torch.float32) if not tensor.is_floating_point() else tensor
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self) -> "Boxes":
        return Boxes(self.tensor.clone())

    def to(self, *args, **kwargs) -> "Boxes":
        return Boxes(self.tensor.to(*args, **kwargs))

    def area(self) -> torch.Tensor:
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size: Tuple[int, int]) -> None:
        h, w = box_size
        x1 = self.tensor[:, 0].clamp(min=0, max=w)
        y1 = self.tensor[:, 1].clamp(min=0, max=h)
        x2 = self.tensor[:, 2].clamp(min=0, max=w)
        y2 = self.tensor[:, 3].clamp(min=0, max=h)
        self.tensor = torch.stack((x1, y1, x2, y2), dim=-1)

    def nonempty(self, threshold: float = 0.0) -> torch.Tensor:
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def inside_box(self, box_size: Tuple[int, int], boundary_threshold: int = 0) -> torch.Tensor:
        h, w = box_size
        b = self.tensor
        return (
            (b[:, 0] >= -boundary_threshold)
            & (b[:, 1] >= -boundary_threshold)
            & (b[:, 2] < w + boundary_threshold)
            & (b[:, 3] < h + boundary_threshold)
        )

    def scale(self, scale_x: float, scale_y: float) -> None:
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    def __getitem__(self, item) -> "Boxes":
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2, "Indexing on Boxes with {} failed".format(item)
        return Boxes(b)

    def __len__(self) -> int:
        return self.tensor.shape[0]

    def __repr__(self) -> str:
        return "Boxes(" + str(self.tensor) + ")"

    @property
    def device(self):
        return self.tensor.device

    @classmethod
    def cat(cls, boxes_list: List["Boxes"]) -> "Boxes":
        assert isinstance(boxes_list, (list, tuple))
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        assert all(isinstance(b, Boxes) for b in boxes_list)
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    def __iter__(self):
        yield from self.tensor


def pairwise_intersection(b1: Boxes, b2: Boxes) -> torch.Tensor:
    a, b = b1.tensor, b2.tensor
    wh = torch.min(a[:, None, 2:], b[:, 2:]) - torch.max(a[:, None, :2], b[:, :2])
    wh.clamp_(min=0)
    return wh.prod(dim=2)


def pairwise_iou(b1: Boxes, b2: Boxes) -> torch.Tensor:
    """IoU matrix [len(b1), len(b2)]; 0 where the intersection is empty."""
    area1, area2 = b1.area(), b2.area()
    inter = pairwise_intersection(b1, b2)
    return torch.where(
        inter > 0,
        inter / (area1[:, None] + area2 - inter),
        torch.zeros(1, dtype=inter.dtype, device=inter.device),
    )


class Instances:
    """Attribute bag whose fields all share one length (detectron2.structures.Instances)."""

    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        self._image_size = image_size
        self._fields: Dict[str, Any] = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self) -> Tuple[int, int]:
        return self._image_size

    def __setattr__(self, name: str, val: Any) -> None:
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name: str) -> Any:
        if name == "_fields" or name not in self._fields:
            raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))
        return self._fields[name]

    def set(self, name: str, value: Any) -> None:
        data_len = len(value)
        if len(self._fields):
            assert len(self) == data_len, "Adding a field of length {} to a Instances of length {}".format(
                data_len, len(self)
            )
        self._fields[name] = value

    def has(self, name: str) -> bool:
        return name in self._fields

    def remove(self, name: str) -> None:
        del self._fields[name]

    def get(self, name: str) -> Any:
        return self._fields[name]

    def get_fields(self) -> Dict[str, Any]:
        return self._fields

    def to(self, *args: Any, **kwargs: Any) -> "Instances":
        ret = type(self)(self._image_size)
        for k, v in self._fields.items():
            if hasattr(v, "to"):
                v = v.to(*args, **kwargs)
            ret._fields[k] = v
        return ret

    def __getitem__(self, item) -> "Instances":
        if type(item) == int:
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = type(self)(self._image_size)
        for k, v in self._fields.items():
            ret._fields[k] = v[item]
        return ret

    def __len__(self) -> int:
        for v in self._fields.values():
            return v.__len__()
        raise NotImplementedError("Empty Instances does not support __len__!")

    def __iter__(self):
        raise NotImplementedError("`Instances` object is not iterable!")

    @staticmethod
    def cat(instance_lists: List["Instances"]) -> "Instances":
        assert all(isinstance(i, Instances) for i in instance_lists)
        assert len(instance_lists) > 0
        if len(instance_lists) == 1:
            return instance_lists[0]
        image_size = instance_lists[0].image_size
        for i in instance_lists[1:]:
            assert i.image_size == image_size
        ret = type(instance_lists[0])(image_size)
        for k in instance_lists[0]._fields.keys():
            values = [i.get(k) for i in instance_lists]
            v0 = values[0]
            if isinstance(v0, torch.Tensor):
                values = torch.cat(values, dim=0)
            elif isinstance(v0, list):
                values = list(itertools.chain(*values))
            elif hasattr(type(v0), "cat"):
                values = type(v0).cat(values)
            else:
                raise ValueError("Unsupported type {} for concatenation".format(type(v0)))
            ret._fields[k] = values
        return ret

    def __str__(self) -> str:
        s = self.__class__.__name__ + "("
        s += "num_instances={}, ".format(len(self) if self._fields else 0)
        s += "image_height={}, image_width={}, ".format(*self._image_size)
        s += "fields=[{}])".format(", ".join(f"{k}: {v}" for k, v in self._fields.items()))
        return s

    __repr__ = __str__


class ImageList:
    """Batched, zero-padded images + true sizes (clip_rcnn.py:297)."""

    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self) -> int:
        return len(self.image_sizes)

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def from_tensors(tensors: List[torch.Tensor], size_divisibility: int = 0, pad_value: float = 0.0) -> "ImageList":
        assert len(tensors) > 0
        image_sizes = [(im.shape[-2], im.shape[-1]) for im in tensors]
        max_h = max(s[0] for s in image_sizes)
        max_w = max(s[1] for s in image_sizes)
        if size_divisibility > 1:
            d = size_divisibility
            max_h = (max_h + d - 1) // d * d
            max_w = (max_w + d - 1) // d * d
        batch = tensors[0].new_full((len(tensors),) + tuple(tensors[0].shape[:-2]) + (max_h, max_w), pad_value)
        for img, pad_img in zip(tensors, batch):
            pad_img[..., : img.shape[-2], : img.shape[-1]].copy_(img)
        return ImageList(batch.contiguous(), image_sizes)


class ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride

    def __repr__(self):
        return f"ShapeSpec(channels={self.channels}, height={self.height}, width={self.width}, stride={self.stride})"


def nonzero_tuple(x: torch.Tensor):
    if x.dim() == 0:
        return x.unsqueeze(0).nonzero().unbind(1)
    return x.nonzero().unbind(1)


def cat(tensors: List[torch.Tensor], dim: int = 0) -> torch.Tensor:
    assert isinstance(tensors, (list, tuple))
    if len(tensors) == 1:
        return tensors[0]
    return torch.cat(tensors, dim)


# --------------------------------------------------------------------------- #
# Matcher / samplers (clip_roi_heads.py:126-130,316,363; rpn.py:160,182,242)
# --------------------------------------------------------------------------- #
class Matcher:
    """Assign each prediction (column) to the gt (row) of highest quality, then label by threshold band."""

    def __init__(self, thresholds: Sequence[float], labels: Sequence[int], allow_low_quality_matches: bool = False):
        thresholds = list(thresholds)
        assert thresholds[0] > 0
        thresholds.insert(0, -float("inf"))
        thresholds.append(float("inf"))
        assert all(lo <= hi for lo, hi in zip(thresholds[:-1], thresholds[1:]))
        assert all(l in (-1, 0, 1) for l in labels)
        assert len(labels) == len(thresholds) - 1
        self.thresholds = thresholds
        self.labels = list(labels)
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, match_quality_matrix: torch.Tensor):
        assert match_quality_matrix.dim() == 2
        if match_quality_matrix.numel() == 0:
            n = match_quality_matrix.size(1)
            default_matches = match_quality_matrix.new_full((n,), 0, dtype=torch.int64)
            default_labels = match_quality_matrix.new_full((n,), self.labels[0], dtype=torch.int8)
            return default_matches, default_labels
        assert torch.all(match_quality_matrix >= 0)
        matched_vals, matches = match_quality_matrix.max(dim=0)
        match_labels = matches.new_full(matches.size(), 1, dtype=torch.int8)
        for l, low, high in zip(self.labels, self.thresholds[:-1], self.thresholds[1:]):
            band = (matched_vals >= low) & (matched_vals < high)
            match_labels[band] = l
        if self.allow_low_quality_matches:
            best_per_gt, _ = match_quality_matrix.max(dim=1)
            _, pred_inds = nonzero_tuple(match_quality_matrix == best_per_gt[:, None])
            match_labels[pred_inds] = 1
        return matches, match_labels


def subsample_labels(labels: torch.Tensor, num_samples: int, positive_fraction: float, bg_label: int):
    """Random fg/bg index subsets; two ``torch.randperm`` draws, positives first."""
    positive = nonzero_tuple((labels != -1) & (labels != bg_label))[0]
    negative = nonzero_tuple(labels == bg_label)[0]
    num_pos = int(num_samples * positive_fraction)
    num_pos = min(positive.numel(), num_pos)
    num_neg = num_samples - num_pos
    num_neg = min(negative.numel(), num_neg)
    perm1 = torch.randperm(positive.numel(), device=positive.device)[:num_pos]
    perm2 = torch.randperm(negative.numel(), device=negative.device)[:num_neg]
    return positive[perm1], negative[perm2]


def add_ground_truth_to_proposals(gt, proposals: List[Instances]) -> List[Instances]:
    """Append the gt boxes (objectness logit ~ +23.03) to each image's proposals (clip_roi_heads.py:292,345)."""
    assert gt is not None and len(proposals) == len(gt)
    if len(proposals) == 0:
        return proposals
    out = []
    logit = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))
    for g, p in zip(gt, proposals):
        gt_boxes = g if isinstance(g, Boxes) else g.gt_boxes
        device = p.objectness_logits.device
        gp = Instances(p.image_size)
        gp.proposal_boxes = gt_boxes
        gp.objectness_logits = logit * torch.ones(len(gt_boxes), device=device)
        out.append(Instances.cat([p, gp]))
    return out


# --------------------------------------------------------------------------- #
# Box regression (fast_rcnn.py:297,619,691,729; rpn.py:300-308)
# --------------------------------------------------------------------------- #
_DEFAULT_SCALE_CLAMP = math.log(1000.0 / 16)


class Box2BoxTransform:
    def __init__(self, weights: Tuple[float, float, float, float], scale_clamp: float = _DEFAULT_SCALE_CLAMP):
        self.weights = tuple(weights)
        self.scale_clamp = scale_clamp

    def get_deltas(self, src_boxes: torch.Tensor, target_boxes: torch.Tensor) -> torch.Tensor:
        sw = src_boxes[:, 2] - src_boxes[:, 0]
        sh = src_boxes[:, 3] - src_boxes[:, 1]
        sx = src_boxes[:, 0] + 0.5 * sw
        sy = src_boxes[:, 1] + 0.5 * sh
        tw = target_boxes[:, 2] - target_boxes[:, 0]
        th = target_boxes[:, 3] - target_boxes[:, 1]
        tx = target_boxes[:, 0] + 0.5 * tw
        ty = target_boxes[:, 1] + 0.5 * th
        wx, wy, ww, wh = self.weights
        dx = wx * (tx - sx) / sw
        dy = wy * (ty - sy) / sh
        dw = ww * torch.log(tw / sw)
        dh = wh * torch.log(th / sh)
        deltas = torch.stack((dx, dy, dw, dh), dim=1)
        assert (sw > 0).all().item(), "Input boxes to Box2BoxTransform are not valid!"
        return deltas

    def apply_deltas(self, deltas: torch.Tensor, boxes: torch.Tensor) -> torch.Tensor:
        deltas = deltas.float() if deltas.dtype != torch.float64 else deltas
        boxes = boxes.to(deltas.dtype)
        w = boxes[:, 2] - boxes[:, 0]
        h = boxes[:, 3] - boxes[:, 1]
        cx = boxes[:, 0] + 0.5 * w
        cy = boxes[:, 1] + 0.5 * h
        wx, wy, ww, wh = self.weights
        dx = deltas[:, 0::4] / wx
        dy = deltas[:, 1::4] / wy
        dw = deltas[:, 2::4] / ww
        dh = deltas[:, 3::4] / wh
        dw = torch.clamp(dw, max=self.scale_clamp)
        dh = torch.clamp(dh, max=self.scale_clamp)
        pcx = dx * w[:, None] + cx[:, None]
        pcy = dy * h[:, None] + cy[:, None]
        pw = torch.exp(dw) * w[:, None]
        ph = torch.exp(dh) * h[:, None]
        x1, y1 = pcx - 0.5 * pw, pcy - 0.5 * ph
        x2, y2 = pcx + 0.5 * pw, pcy + 0.5 * ph
        return torch.stack((x1, y1, x2, y2), dim=-1).reshape(deltas.shape)


def smooth_l1_loss(input: torch.Tensor, target: torch.Tensor, beta: float, reduction: str = "none") -> torch.Tensor:
    """fvcore.nn.smooth_l1_loss; beta < 1e-5 degenerates to L1 (the only mode COIN uses, SMOOTH_L1_BETA 0.0)."""
    if beta < 1e-5:
        loss = torch.abs(input - target)
    else:
        n = torch.abs(input - target)
        loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    if reduction == "mean":
        loss = loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
    elif reduction == "sum":
        loss = loss.sum()
    return loss


def dense_box_regression_loss(
    anchors: List[Boxes],
    box2box_transform: Box2BoxTransform,
    pred_anchor_deltas: List[torch.Tensor],
    gt_boxes: List[torch.Tensor],
    fg_mask: torch.Tensor,
    box_reg_loss_type: str = "smooth_l1",
    smooth_l1_beta: float = 0.0,
) -> torch.Tensor:
    """detectron2 ``_dense_box_regression_loss`` (rpn.py:300-308), smooth_l1 branch only."""
    assert box_reg_loss_type == "smooth_l1"
    a = Boxes.cat(anchors).tensor
    gt_deltas = torch.stack([box2box_transform.get_deltas(a, k) for k in gt_boxes])
    return smooth_l1_loss(cat(pred_anchor_deltas, dim=1)[fg_mask], gt_deltas[fg_mask], beta=smooth_l1_beta, reduction="sum")


# --------------------------------------------------------------------------- #
# Anchors (rpn.py:64)
# --------------------------------------------------------------------------- #
def generate_cell_anchors(sizes: Sequence[float], aspect_ratios: Sequence[float]) -> torch.Tensor:
    anchors = []
    for size in sizes:
        area = size ** 2.0
        for ar in aspect_ratios:
            w = math.sqrt(area / ar)
            h = ar * w
            anchors.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(anchors, dtype=torch.float32)


class DefaultAnchorGenerator(nn.Module):
    box_dim = 4

    def __init__(self, sizes, aspect_ratios, strides, offset: float = 0.0):
        super().__init__()
        self.strides = list(strides)
        n = len(self.strides)
        sizes = list(sizes) * n if len(sizes) == 1 else list(sizes)
        aspect_ratios = list(aspect_ratios) * n if len(aspect_ratios) == 1 else list(aspect_ratios)
        self.cell_anchors = [generate_cell_anchors(s, a) for s, a in zip(sizes, aspect_ratios)]
        for i, c in enumerate(self.cell_anchors):
            self.register_buffer(f"cell_anchors_{i}", c, persistent=False)
        self.offset = offset

    @property
    def num_anchors(self) -> List[int]:
        return [len(c) for c in self.cell_anchors]

    num_cell_anchors = num_anchors

    def forward(self, features: List[torch.Tensor]) -> List[Boxes]:
        out = []
        for i, (f, stride) in enumerate(zip(features, self.strides)):
            base = getattr(self, f"cell_anchors_{i}")
            gh, gw = f.shape[-2:]
            sx = torch.arange(self.offset * stride, gw * stride, step=stride, dtype=torch.float32, device=base.device)
            sy = torch.arange(self.offset * stride, gh * stride, step=stride, dtype=torch.float32, device=base.device)
            yy, xx = torch.meshgrid(sy, sx, indexing="ij")
            xx, yy = xx.reshape(-1), yy.reshape(-1)
            shifts = torch.stack((xx, yy, xx, yy), dim=1)
            out.append(Boxes((shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4)))
        return out


# --------------------------------------------------------------------------- #
# NMS + proposal selection (rpn.py:113-115; fast_rcnn.py:164)
# --------------------------------------------------------------------------- #
def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Greedy NMS, descending score, suppress IoU > threshold (torchvision.ops.nms)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    b = boxes.detach().cpu().numpy().astype(np.float32)
    order = np.argsort(-scores.detach().cpu().numpy().astype(np.float32), kind="stable")
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    suppressed = np.zeros(len(b), dtype=bool)
    keep = []
    for _i in range(len(order)):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1 :]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        inter = np.maximum(0.0, xx2 - xx1) * np.maximum(0.0, yy2 - yy1)
        iou = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[iou > iou_threshold]] = True
    return torch.as_tensor(np.asarray(keep, dtype=np.int64), device=boxes.device)


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Per-category NMS via the coordinate-offset trick (torchvision.ops.batched_nms)."""
    assert boxes.shape[-1] == 4
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    boxes = boxes.float()
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, iou_threshold)


def find_top_rpn_proposals(
    proposals: List[torch.Tensor],
    pred_objectness_logits: List[torch.Tensor],
    image_sizes: List[Tuple[int, int]],
    nms_thresh: float,
    pre_nms_topk: int,
    post_nms_topk: int,
    min_box_size: float,
    training: bool,
) -> List[Instances]:
    num_images = len(image_sizes)
    device = proposals[0].device
    topk_scores, topk_proposals, level_ids = [], [], []
    batch_idx = torch.arange(num_images, device=device)
    for level_id, (proposals_i, logits_i) in enumerate(zip(proposals, pred_objectness_logits)):
        n_i = min(logits_i.shape[1], pre_nms_topk)
        logits_sorted, idx = logits_i.sort(descending=True, dim=1)
        topk_scores_i = logits_sorted.narrow(1, 0, n_i)
        topk_idx = idx.narrow(1, 0, n_i)
        topk_proposals.append(proposals_i[batch_idx[:, None], topk_idx])
        topk_scores.append(topk_scores_i)
        level_ids.append(torch.full((n_i,), level_id, dtype=torch.int64, device=device))
    topk_scores = cat(topk_scores, dim=1)
    topk_proposals = cat(topk_proposals, dim=1)
    level_ids = cat(level_ids, dim=0)
    results = []
    for n, image_size in enumerate(image_sizes):
        boxes = Boxes(topk_proposals[n])
        scores = topk_scores[n]
        lvl = level_ids
        valid = torch.isfinite(boxes.tensor).all(dim=1) & torch.isfinite(scores)
        if not valid.all():
            if training:
                raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")
            boxes, scores, lvl = boxes[valid], scores[valid], lvl[valid]
        boxes.clip(image_size)
        keep = boxes.nonempty(threshold=min_box_size)
        if keep.sum().item() != len(boxes):
            boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
        keep = batched_nms(boxes.tensor, scores, lvl, nms_thresh)
        keep = keep[:post_nms_topk]
        res = Instances(image_size)
        res.proposal_boxes = boxes[keep]
        res.objectness_logits = scores[keep]
        results.append(res)
    return results


# --------------------------------------------------------------------------- #
# RoIAlign (clip_roi_heads.py:142-147,172-176 -> ROIPooler -> torchvision.ops.roi_align,
# aligned=True, sampling_ratio=0).  Direct numpy loops; float64 accumulation optional.
# --------------------------------------------------------------------------- #
def _bilinear_prepare(h: int, w: int, y: float, x: float):
    """(weights, indices) of the 4 taps or None if the sample is outside [-1,H]x[-1,W]."""
    if y < -1.0 or y > h or x < -1.0 or x > w:
        return None
    if y <= 0:
        y = 0.0
    if x <= 0:
        x = 0.0
    y_low, x_low = int(y), int(x)
    if y_low >= h - 1:
        y_high = y_low = h - 1
        y = float(y_low)
    else:
        y_high = y_low + 1
    if x_low >= w - 1:
        x_high = x_low = w - 1
        x = float(x_low)
    else:
        x_high = x_low + 1
    ly, lx = y - y_low, x - x_low
    hy, hx = 1.0 - ly, 1.0 - lx
    return (hy * hx, hy * lx, ly * hx, ly * lx), (y_low, x_low, y_high, x_high)


def _roi_samples(roi: np.ndarray, spatial_scale: float, ph: int, pw: int, sampling_ratio: int, aligned: bool, dtype):
    """Yield (py, px, y, x, count) for every sample point of one RoI, in the kernel's arithmetic type."""
    T = dtype
    off = T(0.5) if aligned else T(0.0)
    x0 = T(roi[1]) * T(spatial_scale) - off
    y0 = T(roi[2]) * T(spatial_scale) - off
    x1 = T(roi[3]) * T(spatial_scale) - off
    y1 = T(roi[4]) * T(spatial_scale) - off
    rw, rh = x1 - x0, y1 - y0
    if not aligned:
        rw, rh = max(rw, T(1.0)), max(rh, T(1.0))
    bh, bw = T(rh) / T(ph), T(rw) / T(pw)
    gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(rh) / ph))
    gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(rw) / pw))
    count = max(gh * gw, 1)
    return x0, y0, bh, bw, gh, gw, count


def roi_align_forward_np(
    feat: np.ndarray, rois: np.ndarray, output_size: Tuple[int, int], spatial_scale: float, sampling_ratio: int = 0,
    aligned: bool = True,
) -> np.ndarray:
    """feat [N,C,H,W], rois [R,5] (batch_idx,x0,y0,x1,y1) -> [R,C,ph,pw].  Arithmetic in feat.dtype."""
    n, c, h, w = feat.shape
    ph, pw = output_size
    T = feat.dtype.type
    out = np.zeros((rois.shape[0], c, ph, pw), dtype=feat.dtype)
    for r in range(rois.shape[0]):
        b = int(rois[r, 0])
        x0, y0, bh, bw, gh, gw, count = _roi_samples(rois[r], spatial_scale, ph, pw, sampling_ratio, aligned, T)
        fm = feat[b]
        for py in range(ph):
            for px in range(pw):
                acc = np.zeros((c,), dtype=feat.dtype)
                for iy in range(gh):
                    y = y0 + T(py) * bh + (T(iy) + T(0.5)) * bh / T(gh)
                    for ix in range(gw):
                        x = x0 + T(px) * bw + (T(ix) + T(0.5)) * bw / T(gw)
                        prep = _bilinear_prepare(h, w, float(y), float(x))
                        if prep is None:
                            continue
                        (w1, w2, w3, w4), (yl, xl, yh, xh) = prep
                        acc += T(w1) * fm[:, yl, xl] + T(w2) * fm[:, yl, xh] + T(w3) * fm[:, yh, xl] + T(w4) * fm[:, yh, xh]
                out[r, :, py, px] = acc / T(count)
    return out


def roi_align_backward_np(
    grad_out: np.ndarray, rois: np.ndarray, feat_shape: Tuple[int, int, int, int], spatial_scale: float,
    sampling_ratio: int = 0, aligned: bool = True,
) -> np.ndarray:
    """Adjoint of :func:`roi_align_forward_np`: scatter-add of grad_out * w / count."""
    n, c, h, w = feat_shape
    ph, pw = grad_out.shape[-2:]
    T = grad_out.dtype.type
    gin = np.zeros(feat_shape, dtype=grad_out.dtype)
    for r in range(rois.shape[0]):
        b = int(rois[r, 0])
        x0, y0, bh, bw, gh, gw, count = _roi_samples(rois[r], spatial_scale, ph, pw, sampling_ratio, aligned, T)
        for py in range(ph):
            for px in range(pw):
                g = grad_out[r, :, py, px] / T(count)
                for iy in range(gh):
                    y = y0 + T(py) * bh + (T(iy) + T(0.5)) * bh / T(gh)
                    for ix in range(gw):
                        x = x0 + T(px) * bw + (T(ix) + T(0.5)) * bw / T(gw)
                        prep = _bilinear_prepare(h, w, float(y), float(x))
                        if prep is None:
                            continue
                        (w1, w2, w3, w4), (yl, xl, yh, xh) = prep
                        gin[b, :, yl, xl] += T(w1) * g
                        gin[b, :, yl, xh] += T(w2) * g
                        gin[b, :, yh, xl] += T(w3) * g
                        gin[b, :, yh, xh] += T(w4) * g
    return gin


def roi_align_torch(
    feat: torch.Tensor, rois: torch.Tensor, output_size: Tuple[int, int], spatial_scale: float, sampling_ratio: int = 0,
    aligned: bool = True,
) -> torch.Tensor:
    """Vectorised, autograd-capable RoIAlign with the same sample placement as the loops above.

    Used where the per-sample numpy loops are too slow (full model forward/backward on CPU).  It is
    itself checked against :func:`roi_align_forward_np` / ``_backward_np`` in tests/test_oracle_d2.py.
    """
    n, c, h, w = feat.shape
    ph, pw = output_size
    R = rois.shape[0]
    if R == 0:
        return feat.new_zeros((0, c, ph, pw))
    dt = feat.dtype
    rois = rois.to(dt)
    off = 0.5 if aligned else 0.0
    bidx = rois[:, 0].long()
    x0 = rois[:, 1] * spatial_scale - off
    y0 = rois[:, 2] * spatial_scale - off
    x1 = rois[:, 3] * spatial_scale - off
    y1 = rois[:, 4] * spatial_scale - off
    rw, rh = x1 - x0, y1 - y0
    if not aligned:
        rw, rh = rw.clamp(min=1.0), rh.clamp(min=1.0)
    bh, bw = rh / ph, rw / pw
    if sampling_ratio > 0:
        gh = torch.full((R,), sampling_ratio, dtype=torch.long)
        gw = torch.full((R,), sampling_ratio, dtype=torch.long)
    else:
        gh = torch.ceil(rh / ph).long().clamp(min=0)
        gw = torch.ceil(rw / pw).long().clamp(min=0)
    count = (gh * gw).clamp(min=1).to(dt)
    GH, GW = max(int(gh.max()), 1), max(int(gw.max()), 1)
    iy = torch.arange(GH, dtype=dt)
    ix = torch.arange(GW, dtype=dt)
    py = torch.arange(ph, dtype=dt)
    px = torch.arange(pw, dtype=dt)
    ghf, gwf = gh.to(dt).clamp(min=1), gw.to(dt).clamp(min=1)
    # y[r, py, iy], x[r, px, ix]
    y = y0[:, None, None] + py[None, :, None] * bh[:, None, None] + (iy[None, None, :] + 0.5) * bh[:, None, None] / ghf[:, None, None]
    x = x0[:, None, None] + px[None, :, None] * bw[:, None, None] + (ix[None, None, :] + 0.5) * bw[:, None, None] / gwf[:, None, None]
    my = (iy[None, None, :] < gh[:, None, None].to(dt)) & ~((y < -1.0) | (y > h))
    mx = (ix[None, None, :] < gw[:, None, None].to(dt)) & ~((x < -1.0) | (x > w))

    def taps(v, size):
        v = v.clamp(min=0)
        lo = v.floor().long()
        edge = lo >= size - 1
        lo = torch.where(edge, torch.full_like(lo, size - 1), lo)
        hi = torch.where(edge, lo, lo + 1)
        v = torch.where(edge, lo.to(dt), v)
        l = v - lo.to(dt)
        return lo, hi, l, 1.0 - l

    yl, yh, ly, hy = taps(y, h)
    xl, xh, lx, hx = taps(x, w)
    hy, ly = hy * my, ly * my
    hx, lx = hx * mx, lx * mx
    # Separable interpolation expressed as two sparse-free contractions per RoI:
    # Wy[r, ph, H] and Wx[r, pw, W] accumulate the (already 1/count-free) tap weights.
    Wy = feat.new_zeros((R, ph, h))
    Wx = feat.new_zeros((R, pw, w))
    Wy.scatter_add_(2, yl.reshape(R, ph, GH), hy.reshape(R, ph, GH))
    Wy.scatter_add_(2, yh.reshape(R, ph, GH), ly.reshape(R, ph, GH))
    Wx.scatter_add_(2, xl.reshape(R, pw, GW), hx.reshape(R, pw, GW))
    Wx.scatter_add_(2, xh.reshape(R, pw, GW), lx.reshape(R, pw, GW))
    # Contract per image (no [R,C,H,W] gather) and in RoI chunks to bound the [r,C,ph,W] intermediate.
    out = feat.new_zeros((R, c, ph, pw))
    for b in bidx.unique().tolist():
        sel = (bidx == b).nonzero()[:, 0]
        for chunk in sel.split(64):
            t = torch.einsum("rph,chw->rcpw", Wy[chunk], feat[b])
            out = out.index_add(0, chunk, torch.einsum("rcpw,rqw->rcpq", t, Wx[chunk]))
    return out / count[:, None, None, None]


class ROIPooler(nn.Module):
    """Single-level ROIAlignV2 pooler (the only form COIN instantiates, clip_roi_heads.py:142-147)."""

    def __init__(self, output_size, scales, sampling_ratio, pooler_type="ROIAlignV2"):
        super().__init__()
        if isinstance(output_size, int):
            output_size = (output_size, output_size)
        assert len(scales) == 1, "C4 pooler has one level"
        assert pooler_type in ("ROIAlignV2", "ROIAlign")
        self.output_size = tuple(output_size)
        self.scale = float(scales[0])
        self.sampling_ratio = int(sampling_ratio)
        self.aligned = pooler_type == "ROIAlignV2"

    def forward(self, x: List[torch.Tensor], box_lists: List[Boxes]) -> torch.Tensor:
        assert len(x) == 1
        rois = convert_boxes_to_pooler_format(box_lists)
        return roi_align_torch(x[0], rois, self.output_size, self.scale, self.sampling_ratio, self.aligned)


def convert_boxes_to_pooler_format(box_lists: List[Boxes]) -> torch.Tensor:
    parts = []
    for i, b in enumerate(box_lists):
        t = b.tensor
        parts.append(torch.cat([torch.full((len(t), 1), float(i), dtype=t.dtype, device=t.device), t], dim=1))
    return cat(parts, dim=0) if parts else torch.zeros((0, 5))


# --------------------------------------------------------------------------- #
# Layers: FrozenBatchNorm2d, StandardRPNHead (utils.py:13,270; rpn.py:65)
# --------------------------------------------------------------------------- #
class FrozenBatchNorm2d(nn.Module):
    _version = 3

    def __init__(self, num_features: int, eps: float = 1e-5):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        bias = self.bias - self.running_mean * scale
        return x * scale.reshape(1, -1, 1, 1).to(x.dtype) + bias.reshape(1, -1, 1, 1).to(x.dtype)

    @classmethod
    def convert_frozen_batchnorm(cls, module: nn.Module) -> nn.Module:
        bn_types = (nn.BatchNorm2d, nn.SyncBatchNorm)
        res = module
        if isinstance(module, bn_types):
            res = cls(module.num_features)
            if module.affine:
                res.weight.data = module.weight.data.clone().detach()
                res.bias.data = module.bias.data.clone().detach()
            res.running_mean.data = module.running_mean.data
            res.running_var.data = module.running_var.data
            res.eps = module.eps
        else:
            for name, child in module.named_children():
                new_child = cls.convert_frozen_batchnorm(child)
                if new_child is not child:
                    res.add_module(name, new_child)
        return res


class StandardRPNHead(nn.Module):
    def __init__(self, in_channels: int, num_anchors: int, box_dim: int = 4):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        self.objectness_logits = nn.Conv2d(in_channels, num_anchors, kernel_size=1, stride=1)
        self.anchor_deltas = nn.Conv2d(in_channels, num_anchors * box_dim, kernel_size=1, stride=1)
        for l in [self.conv, self.objectness_logits, self.anchor_deltas]:
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)

    def forward(self, features: List[torch.Tensor]):
        logits, deltas = [], []
        for x in features:
            t = F.relu(self.conv(x))
            logits.append(self.objectness_logits(t))
            deltas.append(self.anchor_deltas(t))
        return logits, deltas


# --------------------------------------------------------------------------- #
# RPN / ROIHeads base classes: only the members DualTeacherRPN (rpn.py:17) and
# OpenVocabularyRes5ROIHeads (clip_roi_heads.py:91) inherit and call.
# --------------------------------------------------------------------------- #
class RPN(nn.Module):
    def __init__(
        self, *, in_features, head, anchor_generator, anchor_matcher, box2box_transform, batch_size_per_image,
        positive_fraction, pre_nms_topk, post_nms_topk, nms_thresh=0.7, min_box_size=0.0,
        anchor_boundary_thresh=-1.0, loss_weight=1.0, box_reg_loss_type="smooth_l1", smooth_l1_beta=0.0,
    ):
        super().__init__()
        self.in_features = in_features
        self.rpn_head = head
        self.anchor_generator = anchor_generator
        self.anchor_matcher = anchor_matcher
        self.box2box_transform = box2box_transform
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.pre_nms_topk = {True: pre_nms_topk[0], False: pre_nms_topk[1]}
        self.post_nms_topk = {True: post_nms_topk[0], False: post_nms_topk[1]}
        self.nms_thresh = nms_thresh
        self.min_box_size = float(min_box_size)
        self.anchor_boundary_thresh = anchor_boundary_thresh
        if isinstance(loss_weight, float):
            loss_weight = {"loss_rpn_cls": loss_weight, "loss_rpn_loc": loss_weight}
        self.loss_weight = loss_weight
        self.box_reg_loss_type = box_reg_loss_type
        self.smooth_l1_beta = smooth_l1_beta

    def _subsample_labels(self, label: torch.Tensor) -> torch.Tensor:
        pos_idx, neg_idx = subsample_labels(label, self.batch_size_per_image, self.positive_fraction, 0)
        label.fill_(-1)
        label.scatter_(0, pos_idx, 1)
        label.scatter_(0, neg_idx, 0)
        return label

    @torch.no_grad()
    def predict_proposals(self, anchors, pred_objectness_logits, pred_anchor_deltas, image_sizes):
        pred_proposals = self._decode_proposals(anchors, pred_anchor_deltas)
        return find_top_rpn_proposals(
            pred_proposals, pred_objectness_logits, image_sizes, self.nms_thresh,
            self.pre_nms_topk[self.training], self.post_nms_topk[self.training], self.min_box_size, self.training,
        )

    def _decode_proposals(self, anchors: List[Boxes], pred_anchor_deltas: List[torch.Tensor]):
        n = pred_anchor_deltas[0].shape[0]
        out = []
        for a, d in zip(anchors, pred_anchor_deltas):
            b = a.tensor.size(1)
            d = d.reshape(-1, b)
            a = a.tensor.unsqueeze(0).expand(n, -1, -1).reshape(-1, b)
            out.append(self.box2box_transform.apply_deltas(d, a).view(n, -1, b))
        return out


class ROIHeads(nn.Module):
    def __init__(self, *, num_classes, batch_size_per_image, positive_fraction, proposal_matcher, proposal_append_gt=True):
        super().__init__()
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.num_classes = num_classes
        self.proposal_matcher = proposal_matcher
        self.proposal_append_gt = proposal_append_gt

    def _sample_proposals(self, matched_idxs: torch.Tensor, matched_labels: torch.Tensor, gt_classes: torch.Tensor):
        has_gt = gt_classes.numel() > 0
        if has_gt:
            gt_classes = gt_classes[matched_idxs]
            gt_classes[matched_labels == 0] = self.num_classes
            gt_classes[matched_labels == -1] = -1
        else:
            gt_classes = torch.zeros_like(matched_idxs) + self.num_classes
        fg, bg = subsample_labels(gt_classes, self.batch_size_per_image, self.positive_fraction, self.num_classes)
        sampled = torch.cat([fg, bg], dim=0)
        return sampled, gt_classes[sampled]


def detector_postprocess(results: Instances, output_height: int, output_width: int) -> Instances:
    """Rescale detections to the original image size, clip, drop empties (clip_rcnn.py:424)."""
    sx, sy = output_width / results.image_size[1], output_height / results.image_size[0]
    results = Instances((output_height, output_width), **results.get_fields())
    boxes = results.pred_boxes if results.has("pred_boxes") else results.proposal_boxes
    boxes.scale(sx, sy)
    boxes.clip(results.image_size)
    return results[boxes.nonempty()]


def get_warmup_factor_at_iter(method: str, it: int, warmup_iters: int, warmup_factor: float) -> float:
    """detectron2 ``_get_warmup_factor_at_iter`` (lr_scheduler.py:51-53 of the reference calls it)."""
    if it >= warmup_iters:
        return 1.0
    if method == "constant":
        return warmup_factor
    if method == "linear":
        alpha = it / warmup_iters
        return warmup_factor * (1 - alpha) + alpha
    raise ValueError("Unknown warmup method: {}".format(method))
